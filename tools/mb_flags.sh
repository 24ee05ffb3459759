#!/bin/bash
# Compiler-flag sweep on the headline kernel (tools/fftlog_microbench.hip), on the GPU box: bash tools/mb_flags.sh
mkdir -p /tmp/mb
i=0
while IFS= read -r flags; do
  ( hipcc --offload-arch=gfx950 -std=c++17 $flags -o /tmp/mb/f$i tools/fftlog_microbench.hip 2>/tmp/mb/f$i.err || echo "build failed: $flags" ) &
  i=$((i+1))
done <<'LIST'
-O3
-O2
-O3 -mllvm -amdgpu-enable-max-ilp-scheduling-strategy=1
-O3 -mllvm -amdgpu-schedule-metric-bias=0
-O3 -mllvm -amdgpu-schedule-metric-bias=100
-O3 -mllvm -amdgpu-use-divergent-register-indexing=1
-O3 -mllvm -enable-post-misched=0
-O3 -mllvm -amdgpu-early-inline-all=true
-O3 -mllvm -amdgpu-enable-power-sched=1
-O3 -fno-unroll-loops
LIST
wait
i=0
while IFS= read -r flags; do
  echo "== $flags"
  [ -x /tmp/mb/f$i ] && /tmp/mb/f$i 100000 20 | tail -1
  i=$((i+1))
done <<'LIST'
-O3
-O2
-O3 -mllvm -amdgpu-enable-max-ilp-scheduling-strategy=1
-O3 -mllvm -amdgpu-schedule-metric-bias=0
-O3 -mllvm -amdgpu-schedule-metric-bias=100
-O3 -mllvm -amdgpu-use-divergent-register-indexing=1
-O3 -mllvm -enable-post-misched=0
-O3 -mllvm -amdgpu-early-inline-all=true
-O3 -mllvm -amdgpu-enable-power-sched=1
-O3 -fno-unroll-loops
LIST
