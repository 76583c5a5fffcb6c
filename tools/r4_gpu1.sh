#!/bin/bash
# round 4, first GPU pass: tail tests + A/B of FRHIP_TAIL on the step
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "tail or bn_block or bn_lean or partial_sum or stem_gemm" > gpurun_out/r4_t1.log 2>&1; tail -15 gpurun_out/r4_t1.log
timeout 1500 python -m pytest tests/test_gpu_model.py -x -q -k "in_launch or full_step_matches or reproducible or residual_units" > gpurun_out/r4_t2.log 2>&1; tail -15 gpurun_out/r4_t2.log
for v in 1 0 1 0; do
  for n in 16; do
    echo "TAIL=$v NRED=$n"; FRHIP_TAIL=$v FRHIP_TAIL_NRED=$n timeout 600 python bench.py --steps 40 --warmup 10 --no-roofline --no-cpu-baseline --no-other-configs 2>gpurun_out/r4_b.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.3f ms' % d['ms_per_step'], d['config'].get('final_loss'))" || tail -5 gpurun_out/r4_b.err
  done
done 2>&1 | tee gpurun_out/r4_ab_tail.log
for n in 1 4 32; do echo "TAIL=1 NRED=$n"; FRHIP_TAIL=1 FRHIP_TAIL_NRED=$n timeout 600 python bench.py --steps 40 --warmup 10 --no-roofline --no-cpu-baseline --no-other-configs 2>gpurun_out/r4_b.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.3f ms' % d['ms_per_step'], d['config'].get('final_loss'))" || tail -5 gpurun_out/r4_b.err; done 2>&1 | tee -a gpurun_out/r4_ab_tail.log
