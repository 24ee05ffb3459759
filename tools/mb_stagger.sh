#!/bin/bash
# Start offsets between the workgroups of a CU in the headline kernel (tools/fftlog_microbench.hip; -DCP_START_STAGGER=n: workgroup b waits
# ((b / DIV) % MOD) x n x s_sleep(64) = n x 2 us before its first pair), on the GPU box: bash tools/mb_stagger.sh
mkdir -p /tmp/mb
i=0
while IFS= read -r flags; do
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 $flags -o /tmp/mb/s$i tools/fftlog_microbench.hip 2>/tmp/mb/s$i.err || echo "build failed: $flags" ) &
  i=$((i+1))
done <<'LIST'

-DCP_START_STAGGER=1 -DCP_STAGGER_DIV=256 -DCP_STAGGER_MOD=2
-DCP_START_STAGGER=2 -DCP_STAGGER_DIV=256 -DCP_STAGGER_MOD=2
-DCP_START_STAGGER=3 -DCP_STAGGER_DIV=256 -DCP_STAGGER_MOD=2
-DCP_START_STAGGER=5 -DCP_STAGGER_DIV=256 -DCP_STAGGER_MOD=2
-DCP_START_STAGGER=2 -DCP_STAGGER_DIV=1 -DCP_STAGGER_MOD=2
-DCP_START_STAGGER=1 -DCP_STAGGER_DIV=1 -DCP_STAGGER_MOD=4
-DCP_START_STAGGER=1 -DCP_STAGGER_DIV=8 -DCP_STAGGER_MOD=4
LIST
wait
for pass in 1 2; do
i=0
while IFS= read -r flags; do
  echo "== ${flags:-none}"
  [ -x /tmp/mb/s$i ] && /tmp/mb/s$i 100000 20 | tail -1
  i=$((i+1))
done <<'LIST'

-DCP_START_STAGGER=1 -DCP_STAGGER_DIV=256 -DCP_STAGGER_MOD=2
-DCP_START_STAGGER=2 -DCP_STAGGER_DIV=256 -DCP_STAGGER_MOD=2
-DCP_START_STAGGER=3 -DCP_STAGGER_DIV=256 -DCP_STAGGER_MOD=2
-DCP_START_STAGGER=5 -DCP_STAGGER_DIV=256 -DCP_STAGGER_MOD=2
-DCP_START_STAGGER=2 -DCP_STAGGER_DIV=1 -DCP_STAGGER_MOD=2
-DCP_START_STAGGER=1 -DCP_STAGGER_DIV=1 -DCP_STAGGER_MOD=4
-DCP_START_STAGGER=1 -DCP_STAGGER_DIV=8 -DCP_STAGGER_MOD=4
LIST
done
