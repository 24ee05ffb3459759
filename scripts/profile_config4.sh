#!/bin/bash
# Usage (on the GPU box, from the repo root): bash scripts/profile_config4.sh <tag>
# Kernel trace of each filter of config 4 on its own (tools/profile_secondary.py 4w / 4b: chunks of bench.CONFIG4_CHUNK vectors);
# results: gpurun_out/<tag>_config4{w,b}_kernel_stats.csv
tag=${1:-r4}
R=$PWD
export TMPDIR=/tmp
cd /tmp
for c in 4w 4b; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c$c -- python3 $R/tools/profile_secondary.py $c > $R/gpurun_out/prof_c$c.log 2>&1
  cp $(find $R/gpurun_out/prof_c$c -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${tag}_config${c}_kernel_stats.csv
  grep "^{" $R/gpurun_out/prof_c$c.log | tail -1 > $R/gpurun_out/${tag}_config${c}.json
  rm -rf $R/gpurun_out/prof_c$c
done
cd $R
