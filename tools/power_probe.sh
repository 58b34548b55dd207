#!/bin/bash
# Sample board power and shader clock while the metric's step runs: tools/power_probe.sh [steps]
# (rocm-smi needs no privileges for reading).  Output: gpurun_out/power_probe.log
steps=${1:-3000}
out=gpurun_out/power_probe.log; mkdir -p gpurun_out; : > $out
python3 bench.py --steps $steps --warmup 50 --lr 0 --no-cpu-baseline > gpurun_out/power_probe_bench.log 2>&1 &
pid=$!
sleep 12   # import + warm-up
for i in $(seq 1 12); do
  rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -E "Power|sclk|mclk|Temperature \(Sensor (junction|edge)" >> $out
  echo "--" >> $out
  sleep 0.25
done
wait $pid
tail -1 gpurun_out/power_probe_bench.log | cut -c1-200
echo "idle:"; sleep 3; rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk"
