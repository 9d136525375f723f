set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "one_wave_per_simd or causal_temporal" 2>&1 | tail -3
B="timeout -k 10 120 python tools/conv_bench.py fwd 100"
WARM=400 $B 8 32 64 64 2>&1 | grep conv
WARM=400 $B 8 16 128 128 2>&1 | grep conv
WARM=400 KSHAPE=1,3,3 $B 8 32 64 64 2>&1 | grep conv
WARM=400 KSHAPE=1,3,3 SHAPE=32,8,8 $B 8 8 256 256 2>&1 | grep conv
WARM=400 KSHAPE=3,1,1 $B 8 32 64 64 2>&1 | grep conv
WARM=400 KSHAPE=3,1,1 SHAPE=32,16,16 $B 8 16 128 128 2>&1 | grep conv
WARM=400 KSHAPE=3,1,1 SHAPE=32,8,8 $B 8 8 256 256 2>&1 | grep conv
