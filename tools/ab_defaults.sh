#!/bin/bash
# Same-box A/B of the default against single switches: N alternations of 100 timed steps each (ms per step).
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
N=${N:-4}
b() { env "$@" timeout 600 python bench.py --steps 100 --warmup 10 --no-roofline --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.3f' % d['ms_per_step'])"; }
for sw in "$@"; do
  printf "%-28s" "$sw"
  for i in $(seq $N); do printf " default %s | %s %s ;" "$(b A=1)" "$sw" "$(b $sw)"; done
  echo
done
