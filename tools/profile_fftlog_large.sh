#!/bin/bash
# kernel times of the large-size FFTLog path:   gpurun -- bash tools/profile_fftlog_large.sh
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/large
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o large -- python3 $GRAFT_REPO_ROOT/tools/bench_fftlog_large.py > $OUT/bench.txt 2>&1
f=$(find $OUT/prof -name '*kernel_stats.csv' | head -1)
cp "$f" $OUT/kernel_stats.csv
head -12 $OUT/kernel_stats.csv | cut -c1-200
cat $OUT/bench.txt | tail -6
