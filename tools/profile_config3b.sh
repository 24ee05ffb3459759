#!/bin/bash
# kernel stats of config 3B (tools/bench_config3b.py, 8 sigma_rz calls of 10 000 tables)
export TMPDIR=/tmp
rm -rf /tmp/p3b; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p3b -- python3 tools/bench_config3b.py 10000 > /tmp/p3b.log 2>&1
tail -1 /tmp/p3b.log
f=$(find /tmp/p3b -name "*kernel_stats.csv" | head -1)
cp $f gpurun_out/r3_config3b_kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel time per sigma_rz call (8 calls): %.2f ms' % (tot / 8 * 1e-6))
for r in rows[:8]:
    print('  %-70s calls %6s  per call %.3f ms' % (r['Name'][:70], r['Calls'], float(r['TotalDurationNs']) / 8 * 1e-6))
PY
