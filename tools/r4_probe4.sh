#!/bin/bash
# Runs ON THE GPU BOX: small-batch latency with the tile-kernel fold on / off and the in-place stream on / off (same box).
OUT=gpurun_out/r4_probe4
mkdir -p $OUT
export PIO_PROBE_BATCHES=1,2,4,8,12,16
echo "== default";                python tools/latency_probe.py 2>&1 | grep "^B=" | tee $OUT/lat_default.txt
echo "== small-family fold off";  PIO_LN_FOLD_SMALL_MIN_ROWS=1000000 python tools/latency_probe.py 2>&1 | grep "^B=" | tee $OUT/lat_nosmall.txt
echo "== in-place off";           PIO_FOLD_INPLACE=0 python tools/latency_probe.py 2>&1 | grep "^B=" | tee $OUT/lat_noinplace.txt
echo "== fold off";               PIO_LN_FOLD=0 python tools/latency_probe.py 2>&1 | grep "^B=" | tee $OUT/lat_nofold.txt
