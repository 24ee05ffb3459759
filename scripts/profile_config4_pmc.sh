#!/bin/bash
# Usage (on the GPU box, from the repo root): bash scripts/profile_config4_pmc.sh <tag>
# Config 4, one filter at a time (tools/profile_secondary.py 4w / 4b: one untimed chunk and bench.CONFIG4_PROFILE_CHUNKS timed ones of bench.CONFIG4_CHUNK vectors): HBM bytes per kernel from FETCH_SIZE and
# WRITE_SIZE in separate passes (program directly after `--`), with the calibration copies of the same session; summary: profiles/<tag>_config4_traffic.json
tag=${1:-r4y}
R=$PWD
out=$R/gpurun_out/prof4pmc_$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
for w in 4w 4b; do
  RUN="python3 $R/tools/profile_secondary.py $w"
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/${w}_fetch -- $RUN > $out/${w}_fetch.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/${w}_write -- $RUN > $out/${w}_write.log 2>&1
done
hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_cal $R/tools/fetch_calibration.hip 2> $out/cal_build.log
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/cal_fetch -- /tmp/fetch_cal > $out/cal_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/cal_write -- /tmp/fetch_cal > $out/cal_write.log 2>&1
cd $R
python3 scripts/summarize_config4_pmc.py $tag
