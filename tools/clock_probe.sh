#!/bin/bash
# GPU clocks and power while the headline step runs (rocm-smi polled beside a resident 64-pair loop)
cd "$GRAFT_REPO_ROOT" || exit 1
timeout -k 10 120 python3 bench.py --steps 400 --warmup 5 --no-extras --no-cpu-baseline --c3-frames 0 > gpurun_out/clk_bench.json 2> gpurun_out/clk_bench.err &
BP=$!
sleep 25
for i in 1 2 3 4 5 6; do
  timeout 10 rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|mclk|fclk|Power|power" | head -8
  echo ---
  sleep 1.5
done
wait $BP
tail -c 300 gpurun_out/clk_bench.json
