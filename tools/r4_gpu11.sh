#!/bin/bash
# round 4, eleventh GPU pass: kernel tests after the table pruning, grid-barrier experiment, switch matrix, round profile
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; mkdir -p gpurun_out
L=$R/stylegan-for-facerec_amd/frhip/lib
( time timeout 1500 python -m pytest tests/test_gpu_kernels.py -x -q ) > gpurun_out/r4_kernels.log 2>&1; tail -4 gpurun_out/r4_kernels.log
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -k "bf16_full_step or full_size_step or batch_512 or bf16_se" > gpurun_out/r4_t6.log 2>&1; tail -3 gpurun_out/r4_t6.log
# --- grid barrier inside the conv2 launch: the same strip kernel with two counter barriers + a distributed
#     finalize behind its epilogue (libfrhip_gridbar.so), against the product kernel, back to back and behind a cache flush
C=strip_256_256_14_fwd_prelu,strip_128_128_28_fwd
{ for rep in 1 2; do
  echo "# product library (warm / cold)"; python tools/kbench.py suite --iters 40 --only $C 2>&1 | grep -v KBENCH; KBENCH_COLD=1 python tools/kbench.py suite --iters 20 --only $C 2>&1 | grep -v KBENCH
  echo "# + two grid barriers and a distributed BatchNorm finalize behind the epilogue (warm / cold)"; KBENCH_GRIDBAR=1 FRHIP_LIB=$L/libfrhip_gridbar.so python tools/kbench.py suite --iters 40 --only $C 2>&1 | grep -v KBENCH; KBENCH_GRIDBAR=1 KBENCH_COLD=1 FRHIP_LIB=$L/libfrhip_gridbar.so python tools/kbench.py suite --iters 20 --only $C 2>&1 | grep -v KBENCH
done; } > gpurun_out/r4_gridbar.txt 2>&1; cat gpurun_out/r4_gridbar.txt
bash tools/switch_matrix.sh > gpurun_out/r4_switch_matrix.txt 2>&1; cat gpurun_out/r4_switch_matrix.txt
bash tools/profile_round.sh r04 > gpurun_out/r4_profile_round.log 2>&1; tail -20 gpurun_out/r4_profile_round.log
