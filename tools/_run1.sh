set -o pipefail
mkdir -p gpurun_out
timeout -k 10 600 python bench.py > gpurun_out/r04_bench_C2.json 2> gpurun_out/r04_bench_C2.err; tail -c 300 gpurun_out/r04_bench_C2.json; echo
timeout -k 10 600 python bench.py --config C4 > gpurun_out/r04_bench_C4.json 2> gpurun_out/r04_bench_C4.err; tail -c 200 gpurun_out/r04_bench_C4.json; echo
timeout -k 10 900 python bench.py --config C5 > gpurun_out/r04_bench_C5.json 2> gpurun_out/r04_bench_C5.err; tail -c 200 gpurun_out/r04_bench_C5.json; echo
timeout -k 10 900 python tools/volume_bench.py > gpurun_out/r04_volume.log 2>&1; grep -v "Warn\|amdgpu\|base dim" gpurun_out/r04_volume.log | tail -4
