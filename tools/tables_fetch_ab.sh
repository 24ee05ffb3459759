#!/bin/bash
# FETCH_SIZE / WRITE_SIZE of cp_tables_rows_direct for two versions of cp_spline.hip in one session (gpurun -- bash tools/tables_fetch_ab.sh <old-commit>)
old=$1
base="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form"
export TMPDIR=/tmp
R=$PWD
cp cosmoprimo_amd/csrc/cp_spline.hip /tmp/cp_spline_new.hip
cp cosmoprimo_amd/csrc/cp_math.h /tmp/cp_math_new.h
for v in new old; do
  if [ $v = old ]; then cp $R/tools/_cp_spline_old.hip cosmoprimo_amd/csrc/cp_spline.hip; else cp /tmp/cp_spline_new.hip cosmoprimo_amd/csrc/cp_spline.hip; fi
  ( cd cosmoprimo_amd/csrc && hipcc $base -c cp_spline.hip -o cp_spline.o && make > /dev/null 2>&1 ) || echo "build failed"
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_t; ( cd /tmp && timeout 600 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_t -- python3 $R/tools/bench_config3b_kernels.py > /tmp/pmc_t.log 2>&1 )
    python3 - $v $c <<'PY'
import csv, glob, sys
tot = n = 0
for f in glob.glob('/tmp/pmc_t/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if 'tables_rows_direct' in row['Kernel_Name'] and row['Counter_Name'] == sys.argv[2]:
            tot += float(row['Counter_Value']); n += 1
print(sys.argv[1], sys.argv[2], 'per dispatch: %.3f GB (raw counter x 1024; FETCH x 2 for the bytes)' % (tot / max(n, 1) * 1024 / 1e9), n, 'dispatches')
PY
  done
done
cp /tmp/cp_spline_new.hip cosmoprimo_amd/csrc/cp_spline.hip
( cd cosmoprimo_amd/csrc && hipcc $base -c cp_spline.hip -o cp_spline.o && make > /dev/null 2>&1 )
