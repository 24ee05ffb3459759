#!/bin/bash
# per-kernel times of one filter of config 4 (5 chunks of 16 384 vectors):   gpurun -- bash tools/trace_config4_filter.sh 4b|4w
which=${1:-4b}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_$which
mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o t -- python3 $GRAFT_REPO_ROOT/tools/profile_secondary.py $which > $OUT/log.txt 2>&1
python3 - <<PY
import csv
rows = list(csv.DictReader(open('$OUT/prof/t_kernel_stats.csv')))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel time %.2f ms over 5 chunks -> %.3f ms per chunk' % (tot / 1e6, tot / 5e6))
for r in rows[:22]:
    print('%-90s calls %5s  per chunk %.3f ms  (%.1f %%)' % (r['Name'][:90], r['Calls'], float(r['TotalDurationNs']) / 5e6, float(r['Percentage'])))
PY
tail -1 $OUT/log.txt | cut -c1-300
