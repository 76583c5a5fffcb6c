#!/bin/bash
# Same-box A/B of engine / library switches: tools/ab_multi.sh OUT REPS "A=1 B=0" "A=0 B=0" ...
# runs bench.py (headline workload, 50 steps / 10 warm-up, no roofline / CPU legs) REPS times per setting, interleaved.
OUT=$1; REPS=$2; shift 2
: > $OUT
for r in $(seq $REPS); do
  for e in "$@"; do
    ms=$(env $e python bench.py --steps 50 --warmup 10 --no-roofline --no-cpu-baseline --no-other-configs ${AB_ARGS} 2>/dev/null | python -c "import sys,json; print(json.loads(sys.stdin.readline())['ms_per_step'])")
    echo "$ms ms  [$e] ${AB_ARGS}" | tee -a $OUT
  done
done
