#!/bin/bash
# Same-box A/B of two builds of the library on config 3B's FFTLog + spline kernel (tools/bench_geospline.py) and on config 3B itself, builds in turn.
#   bash tools/ab_geospline.sh <before.so> [rounds]
before=$1; rounds=${2:-2}
for r in $(seq 1 $rounds); do
  for entry in "before=$before" "after="; do
    name=${entry%%=*}; lib=${entry#*=}
    echo "== $name"; COSMOPRIMO_AMD_LIBRARY=$lib python tools/bench_geospline.py 2>&1 | grep "^geospline, "
    COSMOPRIMO_AMD_LIBRARY=$lib python tools/bench_config3b.py 2>&1 | grep "config 3B"
  done
done
