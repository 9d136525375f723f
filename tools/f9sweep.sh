set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_rccl.py -x -q -k "one_wave_per_simd or rccl or conv" 2>&1 | tail -5
for f9 in 1 0; do
echo "== DIQT_CONV_F9=$f9"
DIQT_CONV_F9=$f9 WARM=100 timeout -k 10 120 python tools/conv_bench.py fwd 20 8 8 256 256 2>&1 | grep conv
DIQT_CONV_F9=$f9 WARM=100 timeout -k 10 120 python tools/conv_bench.py fwd 20 8 8 384 256 2>&1 | grep conv
DIQT_CONV_F9=$f9 WARM=100 timeout -k 10 120 python tools/conv_bench.py fwd 20 8 8 512 256 2>&1 | grep conv
DIQT_CONV_F9=$f9 WARM=100 timeout -k 10 120 python tools/conv_bench.py fwd 20 8 8 256 384 2>&1 | grep conv
done
timeout -k 10 800 python bench.py --steps 20 --warmup 5 --no-extras 2>&1 | tail -1 | cut -c1-1800
