#!/bin/bash
# round 4, seventh GPU pass: remaining model tests + a kernel trace of the current default
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/prof_r4b
timeout 2400 python -m pytest tests/test_gpu_model.py -x -q -k "bn2_backward or configs0 or bench_size or train_driver_runs or resume or freeze" > gpurun_out/r4_t5.log 2>&1; tail -6 gpurun_out/r4_t5.log; grep -h "vs oracle" gpurun_out/r4_t5.log | cut -c1-420
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r4b/two -o r -- python3 $R/bench.py --steps 12 --warmup 4 --no-roofline --no-cpu-baseline --no-other-configs > $R/gpurun_out/prof_r4b/two.log 2>&1
FRHIP_SINGLE_STREAM=1 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r4b/single -o r -- python3 $R/bench.py --steps 12 --warmup 4 --no-roofline --no-cpu-baseline --no-other-configs > $R/gpurun_out/prof_r4b/single.log 2>&1
cd $R
for m in two single; do db=$(find gpurun_out/prof_r4b/$m -name "*.db" | head -1); python tools/trace_gaps.py $db --csv gpurun_out/prof_r4b/kernel_stats_$m.csv --timeline gpurun_out/prof_r4b/timeline_$m.txt > gpurun_out/prof_r4b/gaps_$m.txt 2>&1; head -3 gpurun_out/prof_r4b/gaps_$m.txt; done
find gpurun_out/prof_r4b -name "*.db" -delete
grep -i "linear\|dropout\|permute\|transpose2d\|conv_wgrad_kernel\|topk\|bn_bwd_apply" gpurun_out/prof_r4b/kernel_stats_single.csv | cut -c1-160
