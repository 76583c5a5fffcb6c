#!/bin/bash
# round 4, fifth GPU pass: tests + event-edge schedule A/B
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "bnbwd2 or tail or bn_block or bn_lean" > gpurun_out/r4_t3.log 2>&1; tail -4 gpurun_out/r4_t3.log
timeout 2400 python -m pytest tests/test_gpu_model.py -x -q -k "in_launch or bn2_backward or reproducible or readiness or two_sgd" > gpurun_out/r4_t4.log 2>&1; tail -4 gpurun_out/r4_t4.log
b() { printf "%-76s " "$*"; env "$@" timeout 600 python bench.py --steps 40 --warmup 10 --no-roofline --no-cpu-baseline --no-other-configs 2>gpurun_out/r4_b.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.3f ms  loss %.3f' % (d['ms_per_step'], d['config'].get('final_loss', 0)))" || tail -5 gpurun_out/r4_b.err; }
OLD="FRHIP_MERGE_EDGES=0 FRHIP_WAIT_EVERY=1 FRHIP_WGRAD_SETS=2"
{
b FRHIP_FUSE_BN2=0
b FRHIP_FUSE_BN2=1
b FRHIP_FUSE_BN2=0 $OLD
b FRHIP_FUSE_BN2=1 $OLD
b FRHIP_FUSE_BN2=0 FRHIP_MERGE_EDGES=0
b FRHIP_FUSE_BN2=0 FRHIP_WAIT_EVERY=1 FRHIP_WGRAD_SETS=2
b FRHIP_FUSE_BN2=0 FRHIP_WAIT_EVERY=4 FRHIP_WGRAD_SETS=6
b FRHIP_FUSE_BN2=1 FRHIP_WAIT_EVERY=4 FRHIP_WGRAD_SETS=6
b FRHIP_FUSE_BN2=0 FRHIP_WGRAD_WGS=192
b FRHIP_FUSE_BN2=1 FRHIP_WGRAD_WGS=192
b FRHIP_FUSE_BN2=0 FRHIP_WGRAD_WGS=256
b FRHIP_FUSE_BN2=0
b FRHIP_FUSE_BN2=1
b FRHIP_FUSE_BN2=0 $OLD
} 2>&1 | tee gpurun_out/r4_ab4.log
