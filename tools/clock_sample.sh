#!/bin/bash
# sample GPU clock / power while a bench runs:  bash tools/clock_sample.sh <steps>
mkdir -p gpurun_out
python bench.py --steps ${1:-1500} --warmup 5 --no-cpu-baseline --no-sequential --no-sincnet > gpurun_out/clk_bench.log 2>&1 &
bp=$!
for i in $(seq 1 60); do
  if ! kill -0 $bp 2>/dev/null; then break; fi
  echo "t=$i $(rocm-smi --showclocks --showpower 2>/dev/null | grep -i 'sclk\|mclk\|fclk\|Power' | tr '\n' ' ' | tr -s ' ')" >> gpurun_out/clk_samples.log
  sleep 0.25
done
wait $bp
grep -c . gpurun_out/clk_samples.log
