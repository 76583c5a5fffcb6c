#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "pack_weights" > gpurun_out/r4_t3.log 2>&1; tail -5 gpurun_out/r4_t3.log | cut -c1-300
timeout 1800 python -m pytest tests/test_gpu_model.py -x -q -k "bf16_full_step or reproducible or folded or g5" > gpurun_out/r4_t6.log 2>&1; tail -5 gpurun_out/r4_t6.log | cut -c1-300
b() { printf "%-50s " "$*"; env "$@" timeout 600 python bench.py --steps 40 --warmup 10 --no-roofline --no-cpu-baseline --no-other-configs 2>gpurun_out/r4_b.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.3f ms  loss %.3f' % (d['ms_per_step'], d['config'].get('final_loss', 0)))" || tail -5 gpurun_out/r4_b.err; }
{ b A=1; b FRHIP_PACK64=0; b A=1; b FRHIP_PACK64=0; b A=1;  b FRHIP_PACK64=0; } 2>&1 | tee gpurun_out/r4_ab16.log
