#!/bin/bash
# Dev tool (GPU box): socket power and clocks (rocm-smi) sampled once a second while bench.py runs its timed steps.
# usage: tools/power_probe.sh [bench.py arguments]
python bench.py --steps 400 --warmup 20 --cpu-sample 0 --no-parity --no-extras "$@" > /tmp/power_probe_bench.json 2>/dev/null &
BP=$!
sleep 25
for i in 1 2 3 4 5 6; do
    /opt/rocm/bin/rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -E "Power|sclk|mclk|busy" | tr '\n' ';'
    echo
    sleep 1
done
wait $BP
tail -c 300 /tmp/power_probe_bench.json | grep -o '"value": [0-9.]*' | head -1
