#!/bin/bash
# Runs ON THE GPU BOX (through gpurun): bench + rocprofv3 kernel trace + two separate PMC passes (FETCH_SIZE, WRITE_SIZE).
# usage: tools/profile_round.sh r1      -> writes gpurun_out/prof_<tag>/...
set -o pipefail
TAG=${1:-r1}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
BENCH="bench.py --steps 10 --warmup 3 --cpu-sample 0 --no-parity --no-extras --launch eager"
echo "== bench (full)"; timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err || exit 1
echo "== kernel trace"; timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python $BENCH > $OUT/trace.log 2>&1 || exit 1
echo "== pmc fetch"; timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python $BENCH > $OUT/pmc_fetch.log 2>&1 || exit 1
echo "== pmc write"; timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python $BENCH > $OUT/pmc_write.log 2>&1 || exit 1
# matrix-pipe occupancy per kernel: cycles the MFMA units were busy (SQ) against the cycles the GPU was active (GRBM):
# both in ONE pass (SQ and GRBM slots are independent), no tracing flags besides the counter collection itself
echo "== pmc mfma"; timeout -k 10 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -- python $BENCH > $OUT/pmc_mfma.log 2>&1 || exit 1
python tools/profile_summarize.py $OUT $TAG
