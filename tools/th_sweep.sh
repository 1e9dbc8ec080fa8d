#!/bin/bash
# GPU box: per-image time of the batched bench for several rows-per-wave values, two rounds
for rep in 1 2; do
  for th in ${TH_LIST:-41 57 73 89 113}; do
    FIXCA_TH=$th python bench.py --no-cpu-baseline --no-extras --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['kernel_plan'].split('th=')[1].split()[0], d['roofline']['us_per_image'])"
  done
done
