for flags in "" "-DCP_SIGMA_ABLATE=2" "" "-DCP_SIGMA_ABLATE=2"; do
  bash tools/variant_lib.sh /tmp/cp_sigma_ablate.so "$flags" cp_sigma.hip || continue
  echo "== flags: $flags"; COSMOPRIMO_AMD_LIBRARY=/tmp/cp_sigma_ablate.so python tools/bench_config3_streams.py 2>&1 | grep -E "fused" | tail -1
done
