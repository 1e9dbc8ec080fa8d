#!/bin/bash
# GPU box (through gpurun from the repo root): the round-3 A/B sweeps behind DESIGN.md's tables, all through tools/ab.py
# (variants share one allocation; every table starts with the box's nt-copy yardstick on the same buffers).
#   tools/sweeps.sh rgb8          RGB8 cubic: tile heights, prefetch depth (needs tools/build_variant.sh rd10 -DRDEPTH=10)
#   tools/sweeps.sh single        one image per launch, RGBA8 cubic: tile heights 20..400
#   tools/sweeps.sh move [fmt]    NONE mover: ring row size x prefetch depth x tile height, single image and batch
#                                 fmt = "bpc bpp" (default: all six wide formats)
set -e
OUT=gpurun_out/sweeps; mkdir -p $OUT
case "${1:-}" in
rgb8)
  python tools/ab.py --bpp 3 --rounds 3 HEAD th25,FIXCA_TH=25 th33,FIXCA_TH=33 th41,FIXCA_TH=41 th49,FIXCA_TH=49 th57,FIXCA_TH=57 \
      rd10r22t41@rd10,FIXCA_RING=22,FIXCA_TH=41 oc192,FIXCA_OC=192 HEAD2 2>&1 | tee $OUT/rgb8.log
  python tools/ab.py --bpp 4 --rounds 3 HEAD 2>&1 | tee -a $OUT/rgb8.log ;;
single)
  python tools/ab.py --single --bpc 1 --bpp 4 --interp 2 --images 8 --rounds 3 --reps 8 HEAD --grid "FIXCA_TH=20,24,29,33,41,45,49,53,61,67,80,100,134,200,400" HEAD2 2>&1 | tee $OUT/single.log ;;
move)
  for cfg in ${2:+"$2"} ${2:-"2 4" "4 4" "-8 4" "2 3" "-4 3" "-8 3"}; do set -- $cfg
    python tools/ab.py --single --bpc $1 --bpp $2 --interp 0 --images 6 --rounds 2 --reps 8 HEAD old,FIXCA_NO_MOVE=1 --grid "FIXCA_MOVE_NU=1,2,4;FIXCA_MOVE_DEPTH=6,10,14;FIXCA_TH=64,160,400,800" 2>&1 | tee -a $OUT/move.log
    python tools/ab.py --bpc $1 --bpp $2 --interp 0 --images 6 --rounds 2 --reps 8 HEAD,per=6 --grid "FIXCA_MOVE_NU=1,2,4;FIXCA_TH=32,64,160;per=6" 2>&1 | tee -a $OUT/move.log
  done ;;
*) echo "usage: tools/sweeps.sh rgb8|single|move [\"bpc bpp\"]"; exit 2 ;;
esac
