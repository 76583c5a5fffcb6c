#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 1800 python -m pytest tests/test_gpu_model.py -x -q -k "bf16_full_step or psp or reproducible or resume or train_driver or two_rank or freeze" > gpurun_out/r4_t6.log 2>&1; tail -4 gpurun_out/r4_t6.log | cut -c1-300
b() { printf "%-50s " "$*"; env "$@" timeout 600 python bench.py --steps 40 --warmup 10 --no-roofline --no-cpu-baseline --no-other-configs 2>gpurun_out/r4_b.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.3f ms  loss %.3f' % (d['ms_per_step'], d['config'].get('final_loss', 0)))" || tail -5 gpurun_out/r4_b.err; }
{ b A=1; b FRHIP_STEM_ROWS_LATE=0; b A=1; b FRHIP_STEM_ROWS_LATE=0; b A=1; b FRHIP_STEM_ROWS_LATE=0; } 2>&1 | tee gpurun_out/r4_ab20.log
