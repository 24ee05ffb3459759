#!/bin/bash
# Usage (GPU box, repo root): bash scripts/final_profiles.sh <tag>
# Everything the committed profiles/<tag>_* files come from, in one session on one box: the headline kernel trace + counters
# (scripts/gpu_profile.sh), the HBM traffic of config 3B kernel by kernel (scripts/profile_config3b_pmc.sh), the instruction census of configs 4
# and 5 and the HBM traffic of config 4 kernel by kernel (scripts/profile_config4_pmc.sh) (scripts/profile_config4_valu.sh, tools/census_config5.sh), the per-kernel statistics of the secondary configs and a plain bench line
# (scripts/profile_secondary.sh, scripts/profile_config4.sh).  Afterwards, in the build container: bash scripts/collect_final_profiles.sh <tag>.
tag=${1:-r4y}
bash scripts/gpu_profile.sh $tag
bash scripts/profile_config3b_pmc.sh $tag > gpurun_out/${tag}_config3b_pmc.log 2>&1
bash scripts/profile_config4_valu.sh $tag > gpurun_out/${tag}_config4_valu.log 2>&1
bash tools/census_config5.sh > gpurun_out/${tag}_config5_census.log 2>&1
cp gpurun_out/census_c5/config5_valu.json gpurun_out/${tag}_config5_valu.json
bash scripts/profile_config4.sh $tag
bash scripts/profile_config4_pmc.sh $tag > gpurun_out/${tag}_config4_pmc.log 2>&1
cp profiles/${tag}_config4_traffic.json gpurun_out/${tag}_config4_traffic.json
bash scripts/profile_secondary.sh $tag
ls gpurun_out | grep "^${tag}_"
