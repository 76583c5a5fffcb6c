#!/bin/bash
# round 4, tenth GPU pass: the FULL gpu suite on the current tree (timing it), then the fp32 fixtures under both tile widths
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
( time timeout 3000 python -m pytest tests -x -q -m gpu --durations=25 ) > gpurun_out/r4_full_suite.log 2>&1; tail -45 gpurun_out/r4_full_suite.log
for w in 128 64; do echo "== FRHIP_IGEMM_BN=$w"; FRHIP_IGEMM_BN=$w timeout 900 python -m pytest tests/test_gpu_model.py -q -s -k "full_step_matches or two_sgd or reference_shaped" 2>&1 | grep -E "fp32 vs golden|per-parameter norms|passed|failed|AssertionError" | cut -c1-260; done > gpurun_out/r4_fp32_tile_width_noise2.txt 2>&1; cat gpurun_out/r4_fp32_tile_width_noise2.txt | grep -v "vs ref fp32" | cut -c1-220
