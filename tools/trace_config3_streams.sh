#!/bin/bash
# kernel trace of one sigma_rz call of config 3 (fused kernel by default; BLOCKS=n for the two-stream block walk): start / end of every dispatch
export TMPDIR=/tmp
cat > /tmp/c3t.py <<PY
import sys
sys.path.insert(0, '.')
import torch, bench
import cosmoprimo_amd as cp
dev = torch.device('cuda', 0)
cp.PowerSpectrumInterpolator2D._two_stream_blocks = ${BLOCKS:-0}
bench.RAMP_S = 0.05
bench.config3(cp, torch, dev, reps=3)
PY
rm -rf /tmp/c3trace; rocprofv3 --kernel-trace --output-format csv -d /tmp/c3trace -- python3 /tmp/c3t.py > /tmp/c3t.log 2>&1
f=$(find /tmp/c3trace -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
last = rows[-8:]
t0 = int(last[0]['Start_Timestamp'])
for r in last:
    print('%-40s queue %s  start %8.1f us  end %8.1f us  vgpr %s lds %s grid %s' % (r['Kernel_Name'].split('(')[0][-40:], r['Queue_Id'], (int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - t0) / 1e3, r['VGPR_Count'], r['LDS_Block_Size'], r['Grid_Size_X']))
PY
