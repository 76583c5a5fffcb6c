#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "residual_sum_behind" > gpurun_out/r4_t3.log 2>&1; tail -12 gpurun_out/r4_t3.log | cut -c1-300
timeout 1800 python -m pytest tests/test_gpu_model.py -x -q -k "residual_sums_formed" -s > gpurun_out/r4_t6.log 2>&1; tail -12 gpurun_out/r4_t6.log | cut -c1-300
c() { printf "%-44s " "$*"; env "${@:2}" timeout 600 python bench.py --steps 30 --warmup 8 --no-roofline --no-cpu-baseline --no-other-configs $1 2>gpurun_out/r4_b.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.3f ms  loss %.3f' % (d['ms_per_step'], d['config'].get('final_loss', 0)))" || tail -5 gpurun_out/r4_b.err; }
C3="--model IR_SE_101 --head CosFace --classes 28000 --batch 128"
C4="--model pSp --head ArcFace --classes 28000 --batch 256"
true
