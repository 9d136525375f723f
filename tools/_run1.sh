set -o pipefail
mkdir -p gpurun_out
WD=300 LIMIT=500 bash tools/rehearse_ranks.sh 2 2; echo "rehearse rc=$?"
tail -c 1500 gpurun_out/rehearse/n2.json; echo; tail -5 gpurun_out/rehearse/n2.err | cut -c1-300
