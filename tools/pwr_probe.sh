# power cap and power draw of the register-only MFMA loop vs the conv kernel
rocm-smi --showmaxpower 2>&1 | grep -i "max\|cap" | head -3
(for i in 1 2 3 4 5 6 7 8 9 10 11 12; do ./tools/mfma_peak > /dev/null 2>&1; done) &
PID=$!
for i in 1 2 3 4; do sleep 1; rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|Socket" | tr '\n' ' '; echo; done
wait $PID
./tools/mfma_peak 2>&1 | tail -3
