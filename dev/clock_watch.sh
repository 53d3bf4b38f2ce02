#!/bin/bash
# sample the shader clock / power while the SpMV probe loops (diagnostic)
python dev/spmv_loop.py "$@" &
PID=$!
sleep 14
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|mclk\|fclk\|power" ; sleep 0.5; done
wait $PID
