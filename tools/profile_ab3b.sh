#!/bin/bash
# Usage (GPU box, repo root): bash tools/profile_ab3b.sh <tag>  -- kernel trace of tools/ab_config3b_geospline.py (the three routes of config 3B)
tag=${1:-r4}
R=$PWD
export TMPDIR=/tmp
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ab3b -- python3 $R/tools/ab_config3b_geospline.py > $R/gpurun_out/${tag}_ab3b.log 2>&1
cp $(find $R/gpurun_out/prof_ab3b -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${tag}_ab3b_kernel_stats.csv
rm -rf $R/gpurun_out/prof_ab3b
cd $R
head -12 gpurun_out/${tag}_ab3b_kernel_stats.csv | cut -c1-220
