#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export FIXCA_DEBUG=1
for oc in 240 192 208 224 128 176; do FIXCA_OC=$oc python tools/r2_rgb.py 2>&1 | grep "C=3" | head -1; done
