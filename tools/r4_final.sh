#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the round's final evidence -- profile_round (bench + rocprofv3 trace + PMC passes),
# the other three configs' bench lines at their class defaults, the small-batch latency table.
set -o pipefail
bash tools/profile_round.sh r4 || exit 1
mkdir -p gpurun_out/r4_final
for c in language flow multimodal; do
  echo "== bench $c"; timeout -k 10 400 python bench.py --config $c > gpurun_out/r4_final/$c.json 2> gpurun_out/r4_final/$c.err || exit 1
done
echo "== latency"; PIO_PROBE_BATCHES=1,2,4,8,12,16,32 timeout -k 10 300 python tools/latency_probe.py > gpurun_out/r4_final/latency.txt 2>&1 || exit 1
grep "^B=" gpurun_out/r4_final/latency.txt
