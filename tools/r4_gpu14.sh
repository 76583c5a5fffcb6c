#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for v in 1 0; do
FRHIP_STEM_IMPLICIT=$v timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --kernel-table gpurun_out/r4_kt_stem$v.json > gpurun_out/r4_kt$v.log 2>&1
python - <<PY
import json
d=json.load(open('gpurun_out/r4_kt_stem$v.json'))
print('STEM_IMPLICIT=$v')
for k,v in d.items():
    if 'stem' in k or 'im2col' in k: print('  ', k, v)
PY
done
