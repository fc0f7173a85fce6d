#!/bin/bash
# Runs ON THE GPU BOX: kernel trace of the ImageNet step with the in-place stream off / on, and the weight-image question
# (fp16: 75 MB of packed images shared by the 8 blocks; fp16sd: 604 MB streamed per step).
set -o pipefail
OUT=gpurun_out/r4_probe1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
BENCH="bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-parity --no-extras --launch eager"
for v in 0 1; do
  export PIO_FOLD_INPLACE=$v
  echo "== trace inplace=$v"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace$v -- python $BENCH > $OUT/trace$v.log 2>&1 || exit 1
done
unset PIO_FOLD_INPLACE
for pol in fp16 fp16sd; do
  echo "== ab $pol"; timeout -k 10 300 python tools/ab_env.py PIO_FOLD_INPLACE=0 PIO_FOLD_INPLACE=1 --policy $pol --graph --rounds 5 > $OUT/ab_$pol.log 2>&1 || exit 1
  tail -2 $OUT/ab_$pol.log
done
for v in 0 1; do f=$(ls $OUT/trace$v/*/*kernel_stats.csv | head -1); cp $f $OUT/kernel_stats_inplace$v.csv; done
