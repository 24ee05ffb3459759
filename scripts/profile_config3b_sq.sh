#!/bin/bash
# Usage (GPU box, repo root): bash scripts/profile_config3b_sq.sh <tag>
# What the two kernels of config 3B wait for: shader counters per kernel (two passes, program directly after `--`); summary printed and written to
# gpurun_out/<tag>_config3b_sq.json
tag=${1:-r4}
R=$PWD
out=$R/gpurun_out/prof3bsq_$tag
mkdir -p $out
export TMPDIR=/tmp
cd /tmp
RUN="python3 $R/tools/profile_secondary.py 3b"
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $out/a -- $RUN > $out/a.log 2>&1
timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $out/b -- $RUN > $out/b.log 2>&1
cd $R
python3 - $out $tag <<'PY'
import csv, glob, json, sys, collections
out, tag = sys.argv[1:3]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.defaultdict(int)
for sub in ('a', 'b'):
    for f in glob.glob(out + '/' + sub + '/**/*counter_collection.csv', recursive=True):
        for row in csv.DictReader(open(f)):
            name = row['Kernel_Name'].split('(')[0].replace('void ', '').replace('(anonymous namespace)::', '')
            acc[name][row['Counter_Name']] += float(row['Counter_Value'])
            if row['Counter_Name'] in ('SQ_WAVE_CYCLES', 'SQ_ACTIVE_INST_VALU'): n[(name, row['Counter_Name'])] += 1
res = {}
for name, c in acc.items():
    if 'SQ_WAVE_CYCLES' not in c or c['SQ_WAVE_CYCLES'] < 1e9: continue
    w = c['SQ_WAVE_CYCLES']
    res[name] = {'dispatches': n[(name, 'SQ_WAVE_CYCLES')], 'fractions_of_wave_cycles': {k: round(v / w, 4) for k, v in c.items() if k.startswith(('SQ_WAIT', 'SQ_ACTIVE', 'SQ_INST_CYCLES', 'SQ_VALU_MFMA'))},
                 'instructions_per_wave_cycle': {k: round(v / w, 5) for k, v in c.items() if k.startswith('SQ_INSTS')}, 'busy_over_wave_cycles': round(c.get('SQ_BUSY_CYCLES', 0) / w, 4)}
json.dump(res, open('gpurun_out/%s_config3b_sq.json' % tag, 'w'), indent=1)
print(json.dumps(res, indent=1))
PY
