#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export FIXCA_DEBUG=1
for pad in 0 2000 4000 6000; do for th in 25 33 49; do echo -n "pad=$pad th=$th "; FIXCA_EXP_LDS_PAD=$pad FIXCA_TH=$th python tools/r2_rgb.py 2>&1 | grep "C=3" | head -1; done; done
for ring in 12 14 20 24; do echo -n "ring=$ring "; FIXCA_RING=$ring python tools/r2_rgb.py 2>&1 | grep "C=3" | head -1; done
