#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): the round's final evidence -- profile_round (bench + rocprofv3 trace + PMC passes),
# the other three configs' bench lines, the small-batch latency table.
set -o pipefail
bash tools/profile_round.sh r3 || exit 1
mkdir -p gpurun_out/r3_final
for c in language flow multimodal; do
  echo "== bench $c"; timeout -k 10 400 python bench.py --config $c > gpurun_out/r3_final/$c.json 2> gpurun_out/r3_final/$c.err || exit 1
done
for pol in fp16x2w fp16x2s fp16/fp16x3f; do
  tag=$(echo $pol | tr / _)
  echo "== bench language $pol"; timeout -k 10 400 python bench.py --config language --policy $pol --cpu-sample 0 > gpurun_out/r3_final/language_$tag.json 2> gpurun_out/r3_final/language_$tag.err || exit 1
done
echo "== latency"; PIO_PROBE_BATCHES=1,2,4,8,12,16,32 timeout -k 10 300 python tools/latency_probe.py > gpurun_out/r3_final/latency.txt 2>&1 || exit 1
cat gpurun_out/r3_final/latency.txt | grep "^B="
