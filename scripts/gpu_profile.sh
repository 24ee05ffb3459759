#!/bin/bash
# Usage (on the GPU box, from the repo root): bash scripts/gpu_profile.sh <tag>
# Kernel trace + HBM / LDS counters of the bench command; summaries land in gpurun_out/prof_<tag>/.
tag=${1:-r1}
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary"
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- $BENCH > $out/trace.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- $BENCH > $out/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -- $BENCH > $out/pmc_write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d $out/pmc_sq -- $BENCH > $out/pmc_sq.log 2>&1
find $out -name "*.csv" | head -50
python3 - <<PY
import csv, glob, collections
for f in glob.glob('$out/trace/**/*kernel_stats.csv', recursive=True):
    print(open(f).read()[:3000])
for name in ('pmc_fetch','pmc_write','pmc_sq'):
    for f in glob.glob('$out/%s/**/*counter_collection.csv' % name, recursive=True):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if 'fftlog' in row.get('Kernel_Name',''):
                acc[row['Counter_Name']].append(float(row['Counter_Value']))
        for k, v in acc.items():
            print(name, k, 'n=%d mean=%.6g' % (len(v), sum(v)/len(v)))
PY
