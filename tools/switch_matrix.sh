#!/bin/bash
# Every A/B switch of the product path through 20 bench steps (N = 1): the step must run and the loss must stay finite.
R=$GRAFT_REPO_ROOT; cd $R
run() { printf "%-48s " "$*"; env "$@" python bench.py --steps 20 --warmup 5 --no-roofline --no-cpu-baseline --no-other-configs 2>/tmp/sw.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.3f ms  loss %.3f' % (d['ms_per_step'], d['config']['final_loss']))" || tail -3 /tmp/sw.err; }
run FRHIP_DEFAULT=1
run FRHIP_SINGLE_STREAM=1
run FRHIP_WGRAD_ROLL=0
run FRHIP_WGRAD_ROLL7=0 FRHIP_WGRAD_S2ROLL=0
run FRHIP_WGRAD_S2ROLL56=0 FRHIP_WGRAD_VR=0
run FRHIP_WGRAD_DEFER=0
run FRHIP_NO_DEFER_SLABS=1
run FRHIP_SPLIT_STRIPS=1
run FRHIP_ROLL64=0 FRHIP_S2ROLL=0
run FRHIP_NO_S2_STRIP=1
run FRHIP_NO_STRIP=1
run FRHIP_SLOPE_ON_MAIN=1
run FRHIP_XCD_ORDER=0
run FRHIP_TAIL=1
run FRHIP_TAIL=1 FRHIP_TAIL_NRED=32
run FRHIP_FUSE_BN2=1
run FRHIP_FUSE_BN2=1 FRHIP_TAIL=1
run FRHIP_LINEAR_CM=0
run FRHIP_RES_MOMENTS=0
run FRHIP_STEM_IMPLICIT=1
run FRHIP_IGEMM_BN=64
run FRHIP_RES_MOMENTS_SE=0
run FRHIP_STEM_TWO_PASS=0
run FRHIP_STEM_RECOMPUTE=0
run FRHIP_C1_STREAM=0
run FRHIP_PACK64=0
