#!/bin/bash
# round 4, second GPU pass: BNBWD2 prologue tests + A/B of FRHIP_FUSE_BN2 and of the tail classes
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "bnbwd2 or tail" > gpurun_out/r4_t3.log 2>&1; tail -12 gpurun_out/r4_t3.log
timeout 2400 python -m pytest tests/test_gpu_model.py -x -q -k "in_launch or bn2_backward or bf16_full_step or bench_size or reproducible" > gpurun_out/r4_t4.log 2>&1; tail -12 gpurun_out/r4_t4.log
b() { printf "%-44s " "$*"; env "$@" timeout 600 python bench.py --steps 40 --warmup 10 --no-roofline --no-cpu-baseline --no-other-configs 2>gpurun_out/r4_b.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.3f ms  loss %.3f' % (d['ms_per_step'], d['config'].get('final_loss', 0)))" || tail -5 gpurun_out/r4_b.err; }
{
b FRHIP_FUSE_BN2=1
b FRHIP_FUSE_BN2=0
b FRHIP_FUSE_BN2=1
b FRHIP_FUSE_BN2=0
b FRHIP_TAIL=1 FRHIP_TAIL_MASK=1 FRHIP_TAIL_NRED=32
b FRHIP_TAIL=1 FRHIP_TAIL_MASK=2 FRHIP_TAIL_NRED=32
b FRHIP_TAIL=1 FRHIP_TAIL_MASK=4 FRHIP_TAIL_NRED=32
b FRHIP_TAIL=1 FRHIP_TAIL_MASK=8 FRHIP_TAIL_NRED=32
b FRHIP_TAIL=1 FRHIP_TAIL_MASK=12 FRHIP_TAIL_NRED=32
b FRHIP_TAIL=1 FRHIP_TAIL_MASK=15 FRHIP_TAIL_NRED=64
b FRHIP_FUSE_BN2=1
} 2>&1 | tee gpurun_out/r4_ab2.log
