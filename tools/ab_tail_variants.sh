#!/bin/bash
# cp_wallish_tail: the shipped kernel against another version of cp_wallish_tail.h (tools/variants/*.h), alternately on ONE box; the variant is built
# beside the shipped library (tools/variant_lib.sh).   bash tools/ab_tail_variants.sh tools/variants/tail_one_wave_splice.h
v=$(cd "$(dirname "$1")" && pwd)/$(basename "$1")
bash tools/variant_lib.sh /tmp/cp_tail_variant.so "-DCP_TAIL_HEADER=\"$v\"" cp_dst.hip || exit 1
for round in 1 2 3; do
  echo "shipped: $(python tools/bench_wallish_tail.py 2>/dev/null | tail -1)"
  echo "variant $(basename $v): $(COSMOPRIMO_AMD_LIBRARY=/tmp/cp_tail_variant.so python tools/bench_wallish_tail.py 2>/dev/null | tail -1)"
done
