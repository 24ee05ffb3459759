#!/bin/bash
# Usage (on the GPU box, from the repo root): bash scripts/profile_config3b_pmc.sh <tag>
# Config 3B (10 000 tabulated P(k, z) -> sigma_rz) under rocprofv3: kernel trace, then HBM bytes per kernel from FETCH_SIZE and WRITE_SIZE in
# separate passes (program directly after `--`), with the calibration copies of the same session; summary: profiles/<tag>_config3b_traffic.json
tag=${1:-r4}
R=$PWD
out=$R/gpurun_out/prof3b_$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
RUN="python3 $R/tools/profile_secondary.py 3b"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $RUN > $out/trace.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- $RUN > $out/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- $RUN > $out/pmc_write.log 2>&1
hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_cal $R/tools/fetch_calibration.hip 2> $out/cal_build.log
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/cal_fetch -- /tmp/fetch_cal > $out/cal_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/cal_write -- /tmp/fetch_cal > $out/cal_write.log 2>&1
cd $R
python3 scripts/summarize_config3b_pmc.py $tag
