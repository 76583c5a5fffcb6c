#!/bin/bash
# usage: ab_multi.sh "<bench args>" "ENV1" "ENV2" ...   (each ENV may hold several VAR=val separated by spaces); 2 passes
cd $GRAFT_REPO_ROOT; ARGS=$1; shift
b() { env "$@" timeout 600 python bench.py --steps 60 --warmup 10 --no-roofline --no-cpu-baseline --no-other-configs $ARGS 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.3f' % d['ms_per_step'])"; }
for pass in 1 2; do for e in "$@"; do printf "%-60s %s\n" "$e" "$(b $e)"; done; done
