#!/bin/bash
# LDS bank conflicts and instruction counts of the XOR-swizzled (default) and the padded additive LDS layout of the flagship kernel
export TMPDIR=/tmp
mkdir -p /tmp/mb gpurun_out/pmc_pad
hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/mb/xor tools/fftlog_microbench.hip 2>&1 | grep error &
hipcc --offload-arch=gfx950 -O3 -std=c++17 -DCP_PADDED_LDS=1 -o /tmp/mb/pad tools/fftlog_microbench.hip 2>&1 | grep error &
wait
for v in xor pad; do
  /tmp/mb/$v 100000 20 | head -1
  timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS --output-format csv -d gpurun_out/pmc_pad/$v -- /tmp/mb/$v 100000 5 > gpurun_out/pmc_pad/$v.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for name in ('xor', 'pad'):
    for f in glob.glob('gpurun_out/pmc_pad/%s/**/*counter_collection.csv' % name, recursive=True):
        acc = collections.defaultdict(list)
        for row in csv.DictReader(open(f)):
            if 'fftlog' in row.get('Kernel_Name', ''):
                acc[row['Counter_Name']].append(float(row['Counter_Value']))
        print(name, {k: '%.4g' % (sum(v) / len(v)) for k, v in sorted(acc.items())})
PY
