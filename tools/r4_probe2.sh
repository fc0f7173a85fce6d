#!/bin/bash
# Runs ON THE GPU BOX: the round's new tests, then the four configs' bench lines and the small-batch latency table.
set -o pipefail
OUT=gpurun_out/r4_probe2
mkdir -p $OUT
echo "== new tests"
timeout -k 10 900 python -m pytest tests/test_parity_gpu.py -q -k "other_widths or shipped_stacks or real_projections or kernels_at_shipped" > $OUT/tests_new.log 2>&1; echo "rc=$?"; grep -E "passed|failed" $OUT/tests_new.log | tail -2
timeout -k 10 900 python -m pytest tests/test_models.py -q -k "trained or s32 or b4_s33" > $OUT/tests_models.log 2>&1; echo "rc=$?"; grep -E "passed|failed" $OUT/tests_models.log | tail -2
for c in language flow multimodal; do
  echo "== bench $c"; timeout -k 10 400 python bench.py --config $c --cpu-sample 0 > $OUT/$c.json 2> $OUT/$c.err || { tail -5 $OUT/$c.err; }
done
echo "== latency"; PIO_PROBE_BATCHES=1,2,4,8,12,16 timeout -k 10 300 python tools/latency_probe.py > $OUT/latency.txt 2>&1; grep "^B=" $OUT/latency.txt
echo "== imagenet"; timeout -k 10 400 python bench.py --cpu-sample 0 > $OUT/imagenet.json 2> $OUT/imagenet.err || tail -5 $OUT/imagenet.err
