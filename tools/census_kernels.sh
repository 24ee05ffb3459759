#!/bin/bash
# vector / matrix / LDS wave-instructions per launch of every kernel of one secondary config:   gpurun -- bash tools/census_kernels.sh 3|3b|4w|4b|5
which=${1:-3}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/census_$which
mkdir -p $OUT
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $OUT/p -o c -- python3 $GRAFT_REPO_ROOT/tools/profile_secondary.py $which > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob('$OUT/p/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    acc[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1].get('SQ_INSTS_VALU', [0]))):
    n = len(v['SQ_INSTS_VALU'])
    print('%-72s launches %4d  per launch: VALU %.4g  MFMA %.4g  LDS %.4g  SALU %.4g' % (k, n, sum(v['SQ_INSTS_VALU']) / n, sum(v.get('SQ_INSTS_MFMA', [0])) / n, sum(v['SQ_INSTS_LDS']) / n, sum(v['SQ_INSTS_SALU']) / n))
PY
