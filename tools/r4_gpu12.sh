#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
b() { printf "%-50s " "$*"; env "$@" timeout 600 python bench.py --steps 40 --warmup 10 --no-roofline --no-cpu-baseline --no-other-configs 2>gpurun_out/r4_b.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.3f ms  loss %.3f' % (d['ms_per_step'], d['config'].get('final_loss', 0)))" || tail -5 gpurun_out/r4_b.err; }
{ b A=1; b FRHIP_EDGE2=0; b FRHIP_WGRAD_ORDER=1; b FRHIP_EDGE2=0 FRHIP_WGRAD_ORDER=1; b A=1; b FRHIP_EDGE2=0; b FRHIP_WGRAD_ORDER=1; b FRHIP_WGRAD_WGS=208; b FRHIP_WGRAD_WGS=240; } 2>&1 | tee gpurun_out/r4_ab7.log
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -k "reproducible or readiness" 2>&1 | tail -3
FRHIP_EDGE2=0 timeout 900 python -m pytest tests/test_gpu_model.py -x -q -k "reproducible or readiness" 2>&1 | tail -3
FRHIP_WGRAD_ORDER=1 timeout 900 python -m pytest tests/test_gpu_model.py -x -q -k "reproducible or readiness" 2>&1 | tail -3
