#!/bin/bash
# tools/build_variant.sh NAME [compiler flags ...]
# Builds the library with extra compiler flags (-DRDEPTH=8, -DROLL_NRB=2, -DFIXCA_EXPERIMENTS ...) into
# tools/bin/NAME/libfixca_hip.so (git-ignored; travels to the GPU box) for tools/ab.py and FIXCA_LIB=.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
mkdir -p "$ROOT/tools/bin/$NAME"
make -s -C "$ROOT/gimp-fix-ca_amd/csrc" OUT="$ROOT/tools/bin/$NAME/libfixca_hip.so" EXTRA="$*"
echo "built tools/bin/$NAME/libfixca_hip.so with: $*"
