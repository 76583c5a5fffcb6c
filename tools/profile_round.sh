#!/bin/bash
# Regenerates the artefacts kept under profiles/ for one round (run on the GPU box: gpurun -- bash tools/profile_round.sh).
TAG=${1:-r03}
# Output: gpurun_out/prof_$TAG/{bench.json,event_table.json,kernel_stats_{two,single}.csv,gaps_*.txt}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/prof_$TAG
cd $R
# the clocks the chip holds while the step runs: every 100-ms sample of the timed loop + a rocm-smi snapshot
python bench.py --steps 200 --warmup 10 --no-roofline --no-cpu-baseline --no-other-configs --clock-log gpurun_out/prof_$TAG/clocks.txt > gpurun_out/prof_$TAG/bench_200.json 2> /dev/null
( echo "# rocm-smi right after the loop:"; rocm-smi --showclocks --showpower 2>/dev/null | grep -E "GPU\[|sclk|Power" ) >> gpurun_out/prof_$TAG/clocks.txt
python bench.py --steps 50 --warmup 10 --kernel-table gpurun_out/prof_$TAG/event_table.json > gpurun_out/prof_$TAG/bench.json 2> gpurun_out/prof_$TAG/bench.err
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG/two -o r -- python3 $R/bench.py --steps 12 --warmup 4 --no-roofline --no-cpu-baseline --no-other-configs > $R/gpurun_out/prof_$TAG/two.log 2>&1
FRHIP_SINGLE_STREAM=1 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_$TAG/single -o r -- python3 $R/bench.py --steps 12 --warmup 4 --no-roofline --no-cpu-baseline --no-other-configs > $R/gpurun_out/prof_$TAG/single.log 2>&1
cd $R
for m in two single; do db=$(find gpurun_out/prof_$TAG/$m -name "*.db" | head -1); python tools/trace_gaps.py $db --csv gpurun_out/prof_$TAG/kernel_stats_$m.csv --timeline gpurun_out/prof_$TAG/timeline_$m.txt > gpurun_out/prof_$TAG/gaps_$m.txt 2>&1; done
find gpurun_out/prof_$TAG -name "*.db" -delete
tail -3 gpurun_out/prof_$TAG/bench.json | cut -c1-600; head -8 gpurun_out/prof_$TAG/gaps_two.txt; head -8 gpurun_out/prof_$TAG/gaps_single.txt
