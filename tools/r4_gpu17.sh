#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "residual_sum_by_its_consumer or conv3x3_strip" > gpurun_out/r4_t3.log 2>&1; tail -5 gpurun_out/r4_t3.log | cut -c1-300
timeout 1800 python -m pytest tests/test_gpu_model.py -x -q -k "residual_sums_formed or bf16_full_step or bf16_se or bench_size or train_driver or resume" > gpurun_out/r4_t6.log 2>&1; tail -8 gpurun_out/r4_t6.log | cut -c1-300
for v in 1 0; do
FRHIP_RES_MOMENTS=$v timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-configs --kernel-table gpurun_out/r4_kt_res$v.json > gpurun_out/r4_kt$v.log 2>&1
done
python - <<PY
import json
a=json.load(open('gpurun_out/r4_kt_res1.json')); b=json.load(open('gpurun_out/r4_kt_res0.json'))
keys=sorted(set(a)|set(b))
for k in keys:
    va=a.get(k,{}); vb=b.get(k,{})
    if not isinstance(va,dict) or not isinstance(vb,dict): continue
    ma=va.get('ms',0); mb=vb.get('ms',0)
    if abs(ma-mb)>0.004: print('%-50s %3d %.4f   | %3d %.4f   d=%.4f' % (k, va.get('launches',0), ma, vb.get('launches',0), mb, ma-mb))
PY
