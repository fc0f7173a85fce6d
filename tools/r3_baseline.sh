#!/bin/bash
# Round-3 baseline on one GPU box: headline bench, ceiling probe (stamps build), B=1 latency with per-class times,
# the other three configs with the GEMM launch log.
set -o pipefail
mkdir -p gpurun_out
python bench.py > gpurun_out/r3_b0.json 2> gpurun_out/r3_b0.err || exit 1
echo "bench done"; tail -c 400 gpurun_out/r3_b0.json
PIO_LIB_PATH=tools/_abl/libpio_wide_0.so python tools/ceiling_probe.py gpurun_out/r3_ceiling.json > gpurun_out/r3_ceiling.log 2>&1 || { tail -20 gpurun_out/r3_ceiling.log; exit 1; }
echo "ceiling done"
python tools/latency_probe.py > gpurun_out/r3_latency.log 2>&1 || { tail -5 gpurun_out/r3_latency.log; exit 1; }
echo "latency done"
PIO_GEMM_LOG=1 python bench.py --batch 1 --steps 2 --warmup 1 --cpu-sample 0 --no-parity --no-extras > gpurun_out/r3_b1.json 2> gpurun_out/r3_b1.shapes || exit 1
for c in language flow multimodal; do
  PIO_GEMM_LOG=1 python bench.py --config $c --cpu-sample 0 --steps 2 --warmup 1 > gpurun_out/r3_cfg0_$c.json 2> gpurun_out/r3_cfg0_$c.shapes || exit 1
  echo "$c done"
done
