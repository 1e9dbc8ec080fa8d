"""GPU box: the cold call (a fresh fixca_cli process on a 6000x4000 RGBA PAM, what a PDB call costs) with its phases,
default (pageable images) and --pinned, N runs each.  bench.py's cold_call_ms leg is the same measurement."""
import os, sys, subprocess, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gimp-fix-ca_amd"))
from fixca import synth
W, H = 6000, 4000
tmp = os.environ.get("TMPDIR", "/tmp")
src, dst = os.path.join(tmp, "cold_in.pam"), os.path.join(tmp, "cold_out.pam")
with open(src, "wb") as f:
    f.write(b"P7\nWIDTH %d\nHEIGHT %d\nDEPTH 4\nMAXVAL 255\nTUPLTYPE RGB_ALPHA\nENDHDR\n" % (W, H))
    f.write(synth.noise_bytes(0xF1CA1000, W * H * 4).tobytes())
cli = os.path.join(ROOT, "gimp-fix-ca_amd", "fixca_cli")
for extra in [[], ["--pinned"]] + ([sys.argv[1:]] if len(sys.argv) > 1 else []):
    for k in range(int(os.environ.get("COLD_RUNS", "4"))):
        env = dict(os.environ, FIXCA_PHASES="1", FIXCA_T0_NS=str(time.clock_gettime_ns(time.CLOCK_MONOTONIC)))
        t = time.perf_counter()
        r = subprocess.run([cli] + extra + ["--lens-as-float", src, dst, "2.0", "-2.0", "658", "1280", "2"], capture_output=True, env=env)
        dt = (time.perf_counter() - t) * 1e3
        ph = [l.split()[1:] for l in r.stderr.decode().splitlines() if l.startswith("fixca-phase")]
        prev = 0.0
        parts = []
        for name, ms in ph:
            parts.append("%s +%.1f" % (name.split(":")[-1], float(ms) - prev))
            prev = float(ms)
        print("%-10s run %d: %.1f ms total (rc %d) | %s | exit +%.1f" % (" ".join(extra) or "pageable", k, dt, r.returncode, " | ".join(parts), dt - prev), flush=True)
