#!/bin/bash
# round 4, third GPU pass: tests again + two-stream timelines with and without FRHIP_FUSE_BN2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out/prof_r4a
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "bnbwd2 or tail or bn_block or bn_lean" > gpurun_out/r4_t3.log 2>&1; tail -6 gpurun_out/r4_t3.log
timeout 2400 python -m pytest tests/test_gpu_model.py -x -q -k "in_launch or bn2_backward or bf16_full_step or bench_size or reproducible" > gpurun_out/r4_t4.log 2>&1; tail -6 gpurun_out/r4_t4.log
cd /tmp
for f in 1 0; do
  FRHIP_FUSE_BN2=$f rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r4a/fuse$f -o r -- python3 $R/bench.py --steps 12 --warmup 4 --no-roofline --no-cpu-baseline --no-other-configs > $R/gpurun_out/prof_r4a/fuse$f.log 2>&1
done
cd $R
for f in 1 0; do db=$(find gpurun_out/prof_r4a/fuse$f -name "*.db" | head -1); python tools/trace_gaps.py $db --csv gpurun_out/prof_r4a/kernel_stats_fuse$f.csv --timeline gpurun_out/prof_r4a/timeline_fuse$f.txt > gpurun_out/prof_r4a/gaps_fuse$f.txt 2>&1; head -4 gpurun_out/prof_r4a/gaps_fuse$f.txt; done
find gpurun_out/prof_r4a -name "*.db" -delete
