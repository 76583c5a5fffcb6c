#!/bin/bash
# round 4, fourth GPU pass: tail tests again; schedule experiments around FRHIP_FUSE_BN2
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
L=$GRAFT_REPO_ROOT/stylegan-for-facerec_amd/frhip/lib
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "bnbwd2 or tail or bn_block or bn_lean" > gpurun_out/r4_t3.log 2>&1; tail -4 gpurun_out/r4_t3.log
timeout 2400 python -m pytest tests/test_gpu_model.py -x -q -k "in_launch or bn2_backward or reproducible" > gpurun_out/r4_t4.log 2>&1; tail -4 gpurun_out/r4_t4.log
b() { printf "%-60s " "$*"; env "$@" timeout 600 python bench.py --steps 40 --warmup 10 --no-roofline --no-cpu-baseline --no-other-configs 2>gpurun_out/r4_b.err | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.3f ms  loss %.3f' % (d['ms_per_step'], d['config'].get('final_loss', 0)))" || tail -5 gpurun_out/r4_b.err; }
{
b FRHIP_FUSE_BN2=1
b FRHIP_FUSE_BN2=0
b FRHIP_FUSE_BN2=1 FRHIP_LIB=$L/libfrhip_wprio.so
b FRHIP_FUSE_BN2=0 FRHIP_LIB=$L/libfrhip_wprio.so
b FRHIP_FUSE_BN2=1 FRHIP_WGRAD_WGS=256
b FRHIP_FUSE_BN2=1 FRHIP_WGRAD_WGS=240
b FRHIP_FUSE_BN2=1 FRHIP_WGRAD_WGS=192
b FRHIP_FUSE_BN2=1 FRHIP_SINGLE_STREAM=1
b FRHIP_FUSE_BN2=0 FRHIP_SINGLE_STREAM=1
b FRHIP_FUSE_BN2=1 FRHIP_WGRAD_SETS=3
b FRHIP_FUSE_BN2=1
b FRHIP_FUSE_BN2=0
} 2>&1 | tee gpurun_out/r4_ab3.log
C=strip_256_256_14_fwd_prelu,strip_256_256_14_dgrad,strip_256_256_14_dgrad_bnbwd2,strip_128_128_28_dgrad,strip_128_128_28_dgrad_bnbwd2,wgs_256_256_14,wgs_256_256_14_bn,wgs_128_128_28
{ echo "# warm (back-to-back)"; python tools/kbench.py suite --iters 30 --only $C 2>&1 | grep -v KBENCH; echo "# cold (1-GiB write between launches)"; KBENCH_COLD=1 python tools/kbench.py suite --iters 20 --only $C 2>&1 | grep -v KBENCH; } > gpurun_out/r4_kbench.log 2>&1; cat gpurun_out/r4_kbench.log
