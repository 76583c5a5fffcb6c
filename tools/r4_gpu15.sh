#!/bin/bash
# what "cold" means for the 14x14 kernels: L2-cold only (48-MB flush), everything cold (1 GiB), and operands re-read into the
# memory-side cache after a full flush (KBENCH_TOUCH)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
C="wgs_256_256_14,strip_256_256_14_fwd,strip_256_256_14_dgrad,strip_128_128_28_fwd,wgs_128_128_28,strip_512_512_7_fwd"
{
echo "== warm (back-to-back)"; python tools/kbench.py suite --only $C --iters 30 | grep -v KBENCH
for mb in 48 128 320 1024; do echo "== flush $mb MB"; KBENCH_COLD=1 KBENCH_FLUSH_MB=$mb python tools/kbench.py suite --only $C --iters 30 | grep -v KBENCH; done
echo "== full flush, operands read once, 48-MB flush"; KBENCH_COLD=1 KBENCH_TOUCH=1 python tools/kbench.py suite --only $C --iters 30 | grep -v KBENCH
} 2>&1 | tee gpurun_out/r4_cold_levels.txt
