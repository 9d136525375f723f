set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "one_wave_per_simd" 2>&1 | tail -5
for f9 in 1 0; do
echo "== DIQT_CONV_F9=$f9"
DIQT_CONV_F9=$f9 WARM=100 timeout -k 10 120 python tools/conv_bench.py fwd 20 8 16 128 128
DIQT_CONV_F9=$f9 WARM=100 timeout -k 10 120 python tools/conv_bench.py fwd 20 8 16 192 128
DIQT_CONV_F9=$f9 WARM=100 KSHAPE=1,3,3 timeout -k 10 120 python tools/conv_bench.py fwd 20 8 32 64 64
DIQT_CONV_F9=$f9 WARM=100 KSHAPE=1,3,3 SHAPE=32,16,16 timeout -k 10 120 python tools/conv_bench.py fwd 20 8 16 128 128
DIQT_CONV_F9=$f9 WARM=100 KSHAPE=1,3,3 SHAPE=32,8,8 timeout -k 10 120 python tools/conv_bench.py fwd 20 8 8 256 256
done
