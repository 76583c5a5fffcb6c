# two-stream kernel trace of bench.py with the full timeline of one step (tools/trace_gaps.py --timeline)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-tl1}
mkdir -p $O
cd /tmp
${FRHIP_LIB:+env FRHIP_LIB=$FRHIP_LIB} rocprofv3 --kernel-trace --stats -d $O/two -o r -- python3 $R/bench.py --steps 12 --warmup 4 --no-roofline --no-cpu-baseline --no-other-configs > $O/two.log 2>&1
db=$(find $O/two -name "*.db" | head -1); python3 $R/tools/trace_gaps.py $db --csv $O/kernel_stats_two.csv --timeline $O/timeline_two.txt > $O/gaps_two.txt 2>&1
head -6 $O/gaps_two.txt
find $O -name "*.db" -delete
