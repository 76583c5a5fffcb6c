#!/bin/bash
# FRHIP_STRIP_VARIANT A/B on one box: the odd-shape strip instances (data gradients of the stage-entry convolutions).
R=$GRAFT_REPO_ROOT; cd $R
for rep in 1 2; do for v in "$@"; do
  echo "== variant $v (rep $rep)"
  export FRHIP_STRIP_VARIANT=$v
  python tools/kbench.py strip 64 128 56 --mode 1 --pro 0 --epi 3 --iters 50 | tail -1
  python tools/kbench.py strip 128 256 28 --mode 1 --pro 0 --epi 3 --iters 50 | tail -1
  python tools/kbench.py strip 256 512 14 --mode 1 --pro 0 --epi 3 --iters 50 | tail -1
  python tools/kbench.py strip 512 512 7 --mode 0 --pro 2 --epi 1 --iters 50 | tail -1
  python bench.py --steps 100 --warmup 10 --no-roofline --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('step ms', d['ms_per_step'])"
done; done
