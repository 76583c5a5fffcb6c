#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests -x -q -m gpu --durations=15 ) > gpurun_out/r4_full_suite2.log 2>&1; tail -28 gpurun_out/r4_full_suite2.log | cut -c1-200
b() { printf "%-50s " "$*"; env "$@" timeout 600 python bench.py --steps 40 --warmup 10 --no-roofline --no-cpu-baseline --no-other-configs 2>gpurun_out/r4_b.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.3f ms  loss %.3f' % (d['ms_per_step'], d['config'].get('final_loss', 0)))" || tail -5 gpurun_out/r4_b.err; }
{ b A=1; b FRHIP_RES_MOMENTS=0; b FRHIP_STEM_IMPLICIT=0; b A=1; b FRHIP_RES_MOMENTS=0; b FRHIP_STEM_IMPLICIT=0;  } 2>&1 | tee gpurun_out/r4_ab10.log
