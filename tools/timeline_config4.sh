#!/bin/bash
# config 4, one filter: how much of the span of its kernels the device is busy, and with what (tiny framework kernels against the library's own)
#   gpurun -- bash tools/timeline_config4.sh 4b|4w
which=${1:-4b}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/timeline_$which
rm -rf $OUT; mkdir -p $OUT
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/prof -o t -- python3 $GRAFT_REPO_ROOT/tools/profile_secondary.py $which > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob
f = glob.glob('$OUT/prof/**/*kernel_trace.csv', recursive=True)[0]
rows = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f))), key=lambda r: r[0])
# the steady part: the last third of the dispatches
rows = rows[len(rows) * 2 // 3:]
span = rows[-1][1] - rows[0][0]
busy = sum(e - s for s, e, _ in rows)
tiny = [(s, e, n) for s, e, n in rows if 'at::native' in n or 'elementwise' in n]
gaps = sum(max(0, rows[i + 1][0] - rows[i][1]) for i in range(len(rows) - 1))
print('%d dispatches over %.2f ms: busy %.2f ms (%.0f %%), gaps %.2f ms; framework kernels: %d dispatches, %.2f ms' % (
    len(rows), span / 1e6, busy / 1e6, 100. * busy / span, gaps / 1e6, len(tiny), sum(e - s for s, e, _ in tiny) / 1e6))
import collections
acc = collections.Counter()
cnt = collections.Counter()
for s, e, n in rows:
    acc[n[:70]] += e - s
    cnt[n[:70]] += 1
for n, t in acc.most_common(14):
    print('  %-70s %5d  %.3f ms' % (n, cnt[n], t / 1e6))
PY
