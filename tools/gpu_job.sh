#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "se_ or fused_se or tail or residual" > gpurun_out/r4_t3.log 2>&1; tail -5 gpurun_out/r4_t3.log | cut -c1-300
timeout 1800 python -m pytest tests/test_gpu_model.py -x -q -k "bf16_se or reproducible or psp or g6 or in_launch or two_sgd" > gpurun_out/r4_t6.log 2>&1; tail -5 gpurun_out/r4_t6.log | cut -c1-300
bash tools/profile_round.sh r04 > gpurun_out/r4_prof.log 2>&1; tail -20 gpurun_out/r4_prof.log | cut -c1-400
bash tools/profile_config3.sh > gpurun_out/r4_prof_c3.log 2>&1; tail -3 gpurun_out/r4_prof_c3.log | cut -c1-300
