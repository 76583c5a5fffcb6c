#!/bin/bash
# Regenerates the artefacts kept under profiles/ for one round (run on the GPU box: gpurun -- bash tools/profile_round.sh).
# Output: gpurun_out/v6/{bench.json,event_table.json,kernel_stats_{two,single}.csv,gaps_*.txt}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/v6
cd $R
python bench.py --steps 50 --warmup 10 --kernel-table gpurun_out/v6/event_table.json > gpurun_out/v6/bench.json 2> gpurun_out/v6/bench.err
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/v6/two -o r -- python3 $R/bench.py --steps 12 --warmup 4 --no-roofline --no-cpu-baseline > $R/gpurun_out/v6/two.log 2>&1
FRHIP_SINGLE_STREAM=1 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/v6/single -o r -- python3 $R/bench.py --steps 12 --warmup 4 --no-roofline --no-cpu-baseline > $R/gpurun_out/v6/single.log 2>&1
cd $R
for m in two single; do db=$(find gpurun_out/v6/$m -name "*.db" | head -1); python tools/trace_gaps.py $db --csv gpurun_out/v6/kernel_stats_$m.csv > gpurun_out/v6/gaps_$m.txt 2>&1; done
find gpurun_out/v6 -name "*.db" -delete
tail -3 gpurun_out/v6/bench.json | cut -c1-600; head -8 gpurun_out/v6/gaps_two.txt; head -8 gpurun_out/v6/gaps_single.txt
