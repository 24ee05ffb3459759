#!/bin/bash
# Usage (on the GPU box, from the repo root): bash scripts/gpu_profile.sh <tag>
# Kernel trace + HBM / LDS / issue counters of the bench command (separate passes, program directly after `--`) and the calibration of
# FETCH_SIZE / WRITE_SIZE for 8- and 16-byte accesses; summaries land in gpurun_out/prof_<tag>/.
tag=${1:-r3}
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $BENCH > $out/trace.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- $BENCH > $out/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- $BENCH > $out/pmc_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d $out/pmc_sq -- $BENCH > $out/pmc_sq.log 2>&1
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $out/pmc_sq2 -- $BENCH > $out/pmc_sq2.log 2>&1
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_grbm -- $BENCH > $out/pmc_grbm.log 2>&1
hipcc --offload-arch=gfx950 -O3 -o /tmp/fetch_cal tools/fetch_calibration.hip 2> $out/cal_build.log
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/cal_fetch -- /tmp/fetch_cal > $out/cal_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/cal_write -- /tmp/fetch_cal > $out/cal_write.log 2>&1
python3 scripts/summarize_steady.py $tag > $out/steady.log 2>&1   # profiles/<tag>_headline_{kernel_stats.csv,steady.json}
find $out -name "*.csv" | wc -l
tail -2 $out/trace.log
