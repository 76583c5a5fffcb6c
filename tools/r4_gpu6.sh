#!/bin/bash
# round 4, sixth GPU pass: new Linear path + regression tests; A/B against the round-3 tree on the same box
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "linear or flatten_order or bnbwd2 or tail or bn_block or bn_lean" > gpurun_out/r4_t3.log 2>&1; tail -4 gpurun_out/r4_t3.log
timeout 2400 python -m pytest tests/test_gpu_model.py tests/test_gpu_dropout.py -x -q -k "two_sgd or full_step or in_launch or bn2_backward or reproducible or dropout or accuracy or reference_shaped or configs0 or bench_size" > gpurun_out/r4_t4.log 2>&1; tail -6 gpurun_out/r4_t4.log; grep -h "vs oracle" gpurun_out/r4_t4.log | cut -c1-400
b() { printf "%-50s " "$*"; env "$@" timeout 600 python bench.py --steps 40 --warmup 10 --no-roofline --no-cpu-baseline --no-other-configs 2>gpurun_out/r4_b.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.3f ms  loss %.3f' % (d['ms_per_step'], d['config'].get('final_loss', 0)))" || tail -5 gpurun_out/r4_b.err; }
r3() { printf "%-50s " "round-3 tree $*"; (cd _r3 && env "$@" timeout 600 python bench.py --steps 40 --warmup 10 --no-roofline --no-cpu-baseline --no-other-configs 2>../gpurun_out/r4_b3.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.3f ms  loss %.3f' % (d['ms_per_step'], d['config'].get('final_loss', 0)))" || tail -5 ../gpurun_out/r4_b3.err); }
{
r3 A=1
b A=1
b FRHIP_LINEAR_CM=0
r3 A=1
b A=1
b FRHIP_LINEAR_CM=0
b FRHIP_FUSE_BN2=0
} 2>&1 | tee gpurun_out/r4_ab5.log
