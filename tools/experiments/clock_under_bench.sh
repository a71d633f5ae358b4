#!/bin/bash
# Sample rocm-smi (sclk, power, temperature) while bench.py runs: is the whole-net rate clock / power limited?
python3 bench.py --no-cpu-baseline --steps 2000 --warmup 20 > /tmp/bench_long.json 2>/dev/null &
BP=$!
sleep 25   # import + warm-up
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|Power|Temperature \(Sensor (junction|edge)" | tr -s ' ' | head -8
  echo "--"
  sleep 1
done
wait $BP
cut -c55-130 /tmp/bench_long.json
