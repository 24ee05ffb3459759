#!/bin/bash
# average shader clock during the operator kernel and during a pure MFMA loop: GRBM_GUI_ACTIVE cycles / kernel duration
#   gpurun -- bash tools/linop_clock.sh
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/linop_clock
mkdir -p $OUT
hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak $GRAFT_REPO_ROOT/tools/mfma_peak.hip
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/linop -o linop -- python3 $GRAFT_REPO_ROOT/tools/linop_dense_only.py > $OUT/linop.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE --output-format csv -d $OUT/peak -o peak -- /tmp/mfma_peak > $OUT/peak.log 2>&1
python3 - <<PY
import csv, glob, collections
for tag in ('linop', 'peak'):
    f = glob.glob('$OUT/%s/**/*counter_collection.csv' % tag, recursive=True)
    if not f:
        print(tag, 'no counter file'); continue
    rows = list(csv.DictReader(open(f[0])))
    agg = collections.defaultdict(list)
    for r in rows:
        if r['Counter_Name'] != 'GRBM_GUI_ACTIVE': continue
        dur = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
        agg[r['Kernel_Name'][:60]].append((float(r['Counter_Value']), dur))
    for k, v in agg.items():
        v = v[len(v) // 2:]
        cyc = sum(a for a, b in v) / len(v); dur = sum(b for a, b in v) / len(v)
        print('%-62s n=%3d  cycles %.4g  duration %.1f us  -> %.3f GHz' % (k, len(v), cyc, dur / 1e3, cyc / dur))
PY
