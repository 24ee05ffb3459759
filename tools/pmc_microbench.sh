#!/bin/bash
# LDS / issue counters of the P = 16 and P = 8 microbench builds (GPU box): bash tools/pmc_microbench.sh
export TMPDIR=/tmp
mkdir -p /tmp/mb gpurun_out/pmc_mb
for P in 8 16; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -DMB_P=$P -o /tmp/mb/h$P tools/fftlog_microbench.hip 2>&1 | grep error & done
wait
for P in 8 16; do
  timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU --output-format csv -d gpurun_out/pmc_mb/h$P -- /tmp/mb/h$P 100000 5 > gpurun_out/pmc_mb/h$P.log 2>&1
  timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_SALU --output-format csv -d gpurun_out/pmc_mb/h${P}b -- /tmp/mb/h$P 100000 5 > gpurun_out/pmc_mb/h${P}b.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for name in ('h8','h8b','h16','h16b'):
    for f in glob.glob('gpurun_out/pmc_mb/%s/**/*counter_collection.csv' % name, recursive=True):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if 'fftlog' in row.get('Kernel_Name',''):
                acc[row['Counter_Name']].append(float(row['Counter_Value']))
        for k, v in sorted(acc.items()):
            print(name, k, 'n=%d mean=%.6g' % (len(v), sum(v)/len(v)))
PY
tail -3 gpurun_out/pmc_mb/h8b.log
