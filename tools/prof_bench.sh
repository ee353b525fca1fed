#!/bin/bash
# kernel trace of the bench step (run on the GPU box): tools/prof_bench.sh TAG [bench args]
TAG=${1:-x}; shift
REPO=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
(cd $REPO && SWIN_GEMM_TUNE=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" > $OUT/bench.json 2> $OUT/bench.err)
F=$(find $OUT -name "*kernel_stats.csv" | head -1)
cp $F $OUT/kernel_stats.csv
python3 $REPO/tools/prof_summary.py $OUT/kernel_stats.csv 13 60 > $OUT/summary.txt
head -75 $OUT/summary.txt
