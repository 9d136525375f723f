# samples sclk / socket power while tools/conv_bench.py loops (is the conv kernel power- or cycle-limited?)
WARM=2000 python tools/conv_bench.py ${1:-fwd} 60000 > /tmp/cb.log 2>&1 &
PID=$!
for i in $(seq 1 16); do sleep 2; rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|Socket" | tr '\n' ' '; echo; done
wait $PID
tail -1 /tmp/cb.log
