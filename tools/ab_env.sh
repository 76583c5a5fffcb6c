#!/bin/bash
# Same-box A/B of one environment switch on any bench configuration: tools/ab_env.sh "SWITCH=0" N -- <bench.py arguments>
# prints N alternations of (default, switch) ms per step.
cd $GRAFT_REPO_ROOT; SW=$1; N=$2; shift 3
b() { env "$@" timeout 600 python bench.py --steps 60 --warmup 10 --no-roofline --no-cpu-baseline --no-other-configs $ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.3f' % d['ms_per_step'])"; }
ARGS="$*"
printf "%-24s %s\n" "$SW" "$ARGS"
for i in $(seq $N); do printf "  default %s | %s %s\n" "$(b A=1)" "$SW" "$(b $SW)"; done
