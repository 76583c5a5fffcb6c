#!/bin/bash
# Two-stream kernel trace of bench.py with the current tree + environment: tools/trace_two.sh TAG [env assignments]
# -> gpurun_out/TAG/{kernel_stats_two.csv,timeline_two.txt,gaps_two.txt}
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/$TAG
for e in "$@"; do export "$e"; done
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$TAG/two -o r -- python3 $R/bench.py --steps 12 --warmup 4 --no-roofline --no-cpu-baseline --no-other-configs ${BENCH_ARGS} > $R/gpurun_out/$TAG/two.log 2>&1
cd $R
db=$(find gpurun_out/$TAG/two -name "*.db" | head -1)
python tools/trace_gaps.py $db --csv gpurun_out/$TAG/kernel_stats_two.csv --timeline gpurun_out/$TAG/timeline_two.txt > gpurun_out/$TAG/gaps_two.txt 2>&1
find gpurun_out/$TAG -name "*.db" -delete
rm -rf gpurun_out/$TAG/two
head -12 gpurun_out/$TAG/gaps_two.txt
