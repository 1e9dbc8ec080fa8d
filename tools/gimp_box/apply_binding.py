#!/usr/bin/env python3
"""Applies INTEGRATION.md's binding to a checkout of JoesCat/gimp-fix-ca (v4.2):

    apply_binding.py <checkout>/fix-ca.c <repo root>

Edits fix-ca.c in place (a .orig copy is kept):
  1. `#include <fixca_hip.h>` after the libgimp includes;
  2. inside the DEFINITION of fix_ca_region (not its prototype), right after the statement that opens the progress
     bar (gimp_progress_init, before the row buffers are allocated), the call into libfixca_hip.so: when the library
     accepts the call and succeeds, fill the progress bar and return; otherwise fall through to the plug-in's own
     loop, so the PDB result never depends on a GPU being present.
Nothing else is touched.  The anchors are matched by pattern, the script carries no reference text.
"""
import re
import shutil
import sys

BINDING = '''\t/* libfixca_hip.so (MI355X): same arguments, same meaning; anything it refuses or fails falls through */
\tif (fixca_region_host (srcPTR, dstPTR, orig_width, orig_height, bytes, bpc,
\t                       (const struct FixCaParams_ *) params, x1, x2, y1, y2, show_progress) == FIXCA_OK) {
\t\tif (show_progress)
\t\t\tgimp_progress_update (1.0);\t/* the bar the row loop fills step by step */
\t\treturn;
\t}
'''


def patch(text):
    lines = text.split("\n")
    out, i, n = [], 0, len(lines)
    included = bound = in_def = False
    while i < n:
        line = lines[i]
        out.append(line)
        if not included and re.match(r'\s*#\s*include\s*<libgimp/gimpui\.h>', line):
            # the library's FixCaParams has the plug-in's layout; only the entry points and codes are needed
            out.append("#define FIXCA_TYPES_NO_PARAMS 1   /* the plug-in declares FixCaParams itself */")
            out.append("#include <fixca_hip.h>")
            included = True
        # the definition: "static void fix_ca_region (" ... ")" followed by a line that is just "{"; inside it, the
        # statement that calls gimp_progress_init (it may span lines: it ends at the first ";")
        if not bound and not in_def and re.match(r'\s*static\s+void\s+fix_ca_region\s*\(', line):
            j = i
            while j < n and ")" not in lines[j]:
                j += 1
            tail = lines[j].split(")", 1)[1].strip() if j < n else ";"
            if tail == "" and j + 1 < n and lines[j + 1].strip() == "{":
                in_def = True
        elif in_def and not bound and "gimp_progress_init" in line:
            j = i
            while j < n and ";" not in lines[j]:
                j += 1
            out.extend(lines[i + 1:j + 1])
            out.append("")
            out.extend(BINDING.rstrip("\n").split("\n"))
            bound = True
            i = j
        i += 1
    if not (included and bound):
        raise SystemExit("anchors not found (include: %s, definition: %s): is this fix-ca.c v4.2?" % (included, bound))
    return "\n".join(out)


def main():
    path = sys.argv[1]
    text = open(path, encoding="utf-8").read()
    if "fixca_region_host" in text:
        print("already patched:", path)
        return
    shutil.copyfile(path, path + ".orig")
    open(path, "w", encoding="utf-8").write(patch(text))
    print("patched:", path)


if __name__ == "__main__":
    main()
