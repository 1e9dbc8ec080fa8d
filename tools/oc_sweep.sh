#!/bin/bash
for rep in 1 2; do
for oc in ${OC_LIST:-240 224 208}; do
  FIXCA_OC=$oc FIXCA_TH=${TH:-33} python bench.py --no-cpu-baseline --no-extras --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('oc', $oc, d['roofline']['us_per_image'])"
done
done
