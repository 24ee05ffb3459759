#!/bin/bash
# counters of the dense-operator kernel (tools/linop_dense_only.py), one pass per group:   gpurun -- bash tools/linop_pmc.sh <tag>
tag=${1:-a}
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/linop_pmc_$tag
mkdir -p $OUT
i=0
for group in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TD_TD_BUSY" "TCP_PENDING_STALL_CYCLES TCP_TCR_TCP_STALL_CYCLES TCP_TCC_READ_REQ TCP_TA_TCP_STATE_READ" "GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $group --output-format csv -d $OUT/p$i -o p -- python3 $GRAFT_REPO_ROOT/tools/linop_dense_only.py > $OUT/p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in sorted(glob.glob('$OUT/p*/**/*counter_collection.csv', recursive=True)):
    for r in csv.DictReader(open(f)):
        if 'linop_mfma_kernel' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in agg.items():
    v = v[len(v) // 2:]
    print('%-36s %.4g' % (k, sum(v) / len(v)))
PY
