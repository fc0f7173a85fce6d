#!/bin/bash
set -o pipefail
OUT=gpurun_out/r4_probe3
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 0 1; do
  echo "== flow ring128=$v"; PIO_GEMM_RING128=$v timeout -k 10 300 python tools/ab_env.py PIO_X=0 --config flow --graph --rounds 5 > $OUT/flow_ring$v.log 2>&1; tail -1 $OUT/flow_ring$v.log
done
for v in 0 1; do
  echo "== mm ring128=$v"; PIO_GEMM_RING128=$v timeout -k 10 300 python tools/ab_env.py PIO_X=0 --config multimodal --graph --rounds 3 --steps 2 > $OUT/mm_ring$v.log 2>&1; tail -1 $OUT/mm_ring$v.log
done
echo "== flow trace"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_flow -- python bench.py --config flow --steps 5 --warmup 2 --cpu-sample 0 --no-parity --no-extras --launch eager > $OUT/trace_flow.log 2>&1
f=$(ls $OUT/trace_flow/*/*kernel_stats.csv | head -1); cp $f $OUT/flow_kernel_stats.csv; head -16 $OUT/flow_kernel_stats.csv | cut -c1-150
echo "== language trace"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace_lang -- python bench.py --config language --steps 3 --warmup 1 --cpu-sample 0 --no-parity --no-extras --launch eager > $OUT/trace_lang.log 2>&1
f=$(ls $OUT/trace_lang/*/*kernel_stats.csv | head -1); cp $f $OUT/lang_kernel_stats.csv; head -22 $OUT/lang_kernel_stats.csv | cut -c1-150
