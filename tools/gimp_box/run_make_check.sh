#!/bin/bash
# GIMP-box kit: build the reference plug-in with the HIP library bound in and run the
# reference's own `make check` (tests/Makefile.am:13-24 -> md5sum -c tests/test1.md5).
#
#   tools/gimp_box/run_make_check.sh <checkout of JoesCat/gimp-fix-ca v4.2> [work dir]
#
# Needs, on one machine: an MI355X with ROCm, gimp-2.10 + gimptool-2.0 + libgimp2.0-dev, autotools, and
# this repository built (python -c "import __graft_entry__ as g; g.build()").  Cannot run in the build
# container (no GIMP): until it has passed somewhere, parity is "unpinned" wherever this repository
# claims it (oracle/README.md).
#
# What it proves when it passes: the library, called from the real plug-in through GIMP's PDB on the
# reference's own test image and arguments, reproduces the reference's fixture byte for byte.
# FIXCA_EXACT=1 selects the fp64 kernel (bit-identical to the CPU loop; the default LINEAR kernel is
# within +-1 LSB, which an md5 does not forgive).  FIXCA_REQUIRE_GPU=1 makes the run fail instead of
# silently passing on the plug-in's own CPU loop.
set -euo pipefail
REF=${1:?path to a checkout of the reference}
REPO=$(cd "$(dirname "$0")/../.." && pwd)
WORK=${2:-$(mktemp -d)}
for tool in gimp-2.10 gimptool-2.0 autoreconf md5sum; do
  command -v $tool >/dev/null || { echo "missing: $tool"; exit 2; }
done
[ -f "$REPO/gimp-fix-ca_amd/libfixca_hip.so" ] || { echo "build the library first"; exit 2; }
rm -rf "$WORK/src" && mkdir -p "$WORK" && cp -r "$REF" "$WORK/src"
cd "$WORK/src"
python3 "$REPO/tools/gimp_box/apply_binding.py" fix-ca.c "$REPO"
# the test plug-in is the same source re-included (tests/test-fix-ca.c), so it gets the binding too
autoreconf -fi
./configure CPPFLAGS="-I$REPO/include" LDFLAGS="-L$REPO/gimp-fix-ca_amd -Wl,-rpath,$REPO/gimp-fix-ca_amd" LIBS="-lfixca_hip"
make -j"$(nproc)"
echo "== baseline: the reference's own CPU loop (library disabled by hiding the GPU)"
HIP_VISIBLE_DEVICES=-1 make check && echo "CPU loop: tests/test1.md5 PASS" || echo "CPU loop: tests/test1.md5 FAIL (GIMP version / JPEG decoder differs from the fixture's?)"
echo "== the HIP library behind fix_ca_region"
if FIXCA_EXACT=1 make check; then
  echo "HIP library (fp64 kernel): tests/test1.md5 PASS"
else
  echo "HIP library (fp64 kernel): tests/test1.md5 FAIL"; exit 1
fi
