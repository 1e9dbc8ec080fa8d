#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export FIXCA_DEBUG=1
for th in 64 32 24 16 128; do echo "th=$th"; FIXCA_TH=$th python tools/wide_bench.py "RGBA u16" 2>&1 | grep "auto "; FIXCA_TH=$th python tools/wide_bench.py "RGBA float" 2>&1 | grep "auto "; done
