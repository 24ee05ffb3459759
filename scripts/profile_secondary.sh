#!/bin/bash
# Usage (on the GPU box, from the repo root): bash scripts/profile_secondary.sh <tag>
# Kernel trace of each secondary config of bench.py on its own (tools/profile_secondary.py) and one full bench line;
# results land in gpurun_out/<tag>_config{3,4,5}_kernel_stats.csv, <tag>_config{3,4,5}.json and <tag>_bench.json.
tag=${1:-r3}
R=$PWD
export TMPDIR=/tmp
cd /tmp
for c in 3 3b 4 5; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_c$c -- python3 $R/tools/profile_secondary.py $c > $R/gpurun_out/prof_c$c.log 2>&1
  cp $(find $R/gpurun_out/prof_c$c -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${tag}_config${c}_kernel_stats.csv
  grep -v "^[WEI]2026" $R/gpurun_out/prof_c$c.log | grep "^{" | tail -1 > $R/gpurun_out/${tag}_config${c}.json
  rm -rf $R/gpurun_out/prof_c$c $R/gpurun_out/prof_c$c.log
done
cd $R
python3 bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.log
tail -c 1500 gpurun_out/${tag}_bench.json
