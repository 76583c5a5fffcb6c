#!/bin/bash
# Every A/B switch of the product path (README table) through 30 bench steps (N = 1): the step must run, the loss must stay
# finite, and the default must be the fastest.  gpurun -- bash tools/switch_matrix.sh > profiles/rNN_switch_matrix.txt
R=$GRAFT_REPO_ROOT; cd $R
run() { printf "%-48s " "$*"; env "$@" python bench.py --steps 30 --warmup 8 --no-roofline --no-cpu-baseline --no-other-configs 2>/tmp/sw.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.3f ms  loss %.3f' % (d['ms_per_step'], d['config']['final_loss']))" || tail -3 /tmp/sw.err; }
run FRHIP_DEFAULT=1
run FRHIP_SINGLE_STREAM=1
run FRHIP_WGRAD_WGS=224
run FRHIP_WGRAD_ROLL=0
run FRHIP_WGRAD_DEFER=0
run FRHIP_RES_MOMENTS=0
run FRHIP_SPLIT_STRIPS=1
run FRHIP_ROLL64=0
run FRHIP_NO_STEM_GEMM=1
run FRHIP_NO_STRIP=1
run FRHIP_XCD_ORDER=0
run FRHIP_DEFAULT=1
