#!/bin/bash
# A/B of library builds on one box: tools/ab_libs.sh libA.so libB.so ...  (paths relative to frhip/lib/)
# For each: the strip cases of tools/kbench.py suite and two 100-step bench runs, interleaved twice (clocks drift).
R=$GRAFT_REPO_ROOT; cd $R
L=$R/stylegan-for-facerec_amd/frhip/lib
CASES=${CASES:-strip_64_64_112_dgrad,strip_64_64_56_dgrad,strip_128_128_28_fwd,strip_256_256_14_fwd_bn,strip_256_256_14_dgrad,strip_512_512_7_fwd}
for rep in 1 2; do for lib in "$@"; do
  echo "== $lib (rep $rep)"
  FRHIP_LIB=$L/$lib python tools/kbench.py suite --iters 50 --only $CASES 2>&1 | grep -v KBENCH
  FRHIP_LIB=$L/$lib python bench.py --steps 100 --warmup 10 --no-roofline --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('step ms', d['ms_per_step'])"
done; done
