#!/bin/bash
# Usage (build container, repo root, after `gpurun -- bash scripts/final_profiles.sh <tag>`): bash scripts/collect_final_profiles.sh <tag>
# Summaries of gpurun_out/ into the committed profiles/<tag>_* files.
tag=${1:-r4y}
python3 scripts/summarize_profile.py $tag > /dev/null
python3 scripts/summarize_steady.py $tag > /dev/null
python3 scripts/summarize_config3b_pmc.py $tag > /dev/null
for f in config4_valu.json config4_traffic.json config5_valu.json config3.json config3b.json config4.json config5.json bench.json config3_kernel_stats.csv config3b_kernel_stats.csv \
         config4_kernel_stats.csv config5_kernel_stats.csv config4w_kernel_stats.csv config4b_kernel_stats.csv; do
  [ -f gpurun_out/${tag}_$f ] && cp gpurun_out/${tag}_$f profiles/${tag}_$f
done
ls profiles | grep "^${tag}_"
