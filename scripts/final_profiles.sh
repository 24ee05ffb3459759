#!/bin/bash
# Usage (GPU box, repo root): bash scripts/final_profiles.sh <tag>
# Everything the committed profiles/<tag>_* files come from, in one session: the headline kernel trace + counters (scripts/gpu_profile.sh), the
# instruction census of config 4 (scripts/profile_config4_valu.sh), the per-kernel statistics of the secondary configs and a plain bench line
# (scripts/profile_secondary.sh).  Copy gpurun_out/<tag>_* and gpurun_out/prof_<tag>/ summaries into profiles/ afterwards.
tag=${1:-r3z}
bash scripts/gpu_profile.sh $tag
bash scripts/profile_config4_valu.sh $tag
bash scripts/profile_secondary.sh $tag
ls gpurun_out | head -40
