"""Scan gfx950 assembly (hipcc -S --cuda-device-only) for a 12/16-byte buffer or global store whose data
registers are written by a VALU instruction within the next two instructions.

The compiler leaves no wait state there when the store's scalar offset is a register (it counts such a store as two
issue cycles); on MI355X that overwrite reached the store: the last lanes of each group of 16 wrote the new value
(profiles/NOTES.md, k_wcol).  The kernels therefore give every such store a constant scalar offset; this scan is the check.

    python tools/store_hazard_scan.py file.s [...]      exit status 1 if any hit
"""
import re, sys

def regs(tok):
    m = re.match(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'v(\d+)$', tok)
    return {int(m.group(1))} if m else set()

def scan(path):
    lines = [l.rstrip() for l in open(path)]
    name, names = None, {}
    for i, l in enumerate(lines):
        if re.match(r'^_Z\w+:', l):
            name = l.split(':')[0]
        names[i] = name
    ins = [(i, l.strip()) for i, l in enumerate(lines) if l.startswith('\t') and not l.strip().startswith(('.', ';'))]
    hits = []
    for k, (i, t) in enumerate(ins):
        m = re.match(r'(buffer|global)_store_dwordx[34]\s', t)
        if not m:
            continue
        ops = [o.strip() for o in t.split(None, 1)[1].split(',')]
        data = regs(ops[0]) if m.group(1) == 'buffer' else regs(ops[1])
        for d in (1, 2):
            if k + d >= len(ins):
                break
            j, u = ins[k + d]
            if not u.startswith('v_') or ' ' not in u:
                continue
            if regs(u.split(None, 1)[1].split(',')[0].strip()) & data:
                hits.append("%s:%d %s: `%s` then (+%d) `%s`" % (path, i + 1, names[i], t, d, u))
    return hits

if __name__ == "__main__":
    all_hits = [h for f in sys.argv[1:] for h in scan(f)]
    for h in all_hits:
        print(h)
    print("hits:", len(all_hits))
    sys.exit(1 if any("+1" in h for h in all_hits) else 0)
