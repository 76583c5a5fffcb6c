#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
for v in "A=1" "FRHIP_PACK64=0" "FRHIP_IGEMM_BN=64"; do
env $v timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --kernel-table gpurun_out/r4_kt_x.json > gpurun_out/r4_ktx.log 2>&1
python - <<PY
import json
d=json.load(open('gpurun_out/r4_kt_x.json'))
print('$v')
for k,v in d.items():
    if isinstance(v,dict) and ('pack' in k or 'igemm' in k or k.startswith('conv_wgrad<')): print('  ', k, v)
PY
done
