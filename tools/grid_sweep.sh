#!/bin/bash
# GPU box: per-image time of the batched bench over ring sizes and rows per wave
for ring in ${RING_LIST:-16 14 12}; do
  for th in ${TH_LIST:-33 44}; do
    FIXCA_RING=$ring FIXCA_TH=$th python bench.py --no-cpu-baseline --no-extras --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ring', $ring, 'th', $th, d['roofline']['us_per_image'])"
  done
done
