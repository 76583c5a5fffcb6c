#!/bin/bash
# Whole-step A/B of library builds on one box: tools/ab_libs_step.sh "<bench args>" libA.so libB.so ... (relative to frhip/lib/), 2 passes
R=$GRAFT_REPO_ROOT; cd $R; L=$R/stylegan-for-facerec_amd/frhip/lib; ARGS=$1; shift
for rep in 1 2; do for lib in "$@"; do
  printf "%-28s %s\n" $lib "$(FRHIP_LIB=$L/$lib python bench.py --steps 60 --warmup 10 --no-roofline --no-cpu-baseline --no-other-configs $ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.3f' % d['ms_per_step'])")"
done; done
