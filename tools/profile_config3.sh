#!/bin/bash
# Kernel-trace of BASELINE configs[3] (IR-SE-101 + CosFace(28000), bs 128): where its step goes, one stream.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/prof_c3; mkdir -p $O
FRHIP_SINGLE_STREAM=1 rocprofv3 --kernel-trace --stats -d $O/single -o r -- python3 $R/bench.py --model IR_SE_101 --head CosFace --classes 28000 --batch 128 --steps 8 --warmup 4 --no-roofline --no-cpu-baseline --no-other-configs > $O/single.log 2>&1
cd $R
db=$(find $O/single -name "*.db" | head -1); python tools/trace_gaps.py $db --csv $O/kernel_stats_single.csv --timeline $O/timeline_single.txt > $O/gaps_single.txt 2>&1
find $O -name "*.db" -delete
head -4 $O/gaps_single.txt; head -45 $O/kernel_stats_single.csv | cut -c1-150
python bench.py --model IR_SE_101 --head CosFace --classes 28000 --batch 128 --steps 30 --warmup 5 --no-roofline --no-cpu-baseline --no-other-configs 2>/dev/null | cut -c1-300
