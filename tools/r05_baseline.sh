#!/bin/bash
# Round-5 baseline on one box: kbench suite (warm + cold for the strip cases) and two 100-step bench runs.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
python tools/kbench.py suite --iters 50 > gpurun_out/r05_base_kbench.txt 2>&1
KBENCH_COLD=1 python tools/kbench.py suite --iters 20 --only strip_256_256_14_fwd_bn,strip_256_256_14_fwd_prelu,strip_256_256_14_dgrad,strip_128_128_28_fwd,strip_128_128_28_dgrad,strip_512_512_7_fwd,wgs_256_256_14 > gpurun_out/r05_base_kbench_cold.txt 2>&1
for i in 1 2; do python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-other-configs 2>/dev/null > gpurun_out/r05_base_bench_$i.json; done
grep -o '"ms_per_step": [0-9.]*' gpurun_out/r05_base_bench_*.json
cat gpurun_out/r05_base_kbench.txt | grep -v KBENCH
