#!/bin/bash
# A variant of the library BESIDE the shipped one:   bash tools/variant_lib.sh <out.so> "<flags>" <source.hip> [<source.hip> ...]
# compiles the named sources of cosmoprimo_amd/csrc with the extra flags into private objects and links them with the other objects of the tree as they
# are.  cosmoprimo_amd/libcosmoprimo_amd.so and the objects next to the sources are never touched: a failed build, a timeout or Ctrl-C leaves the tree as
# it was.  Use the variant with COSMOPRIMO_AMD_LIBRARY=<out.so> (cosmoprimo_amd/_lib.py).  Exit status 1 when the build fails (callers skip the run).
out=$1; flags=$2; shift 2
csrc="$(cd "$(dirname "$0")/../cosmoprimo_amd/csrc" && pwd)"
tmp=$(mktemp -d /tmp/cp_variant.XXXXXX)
trap 'rm -rf "$tmp"' EXIT
base="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form"
( cd "$csrc" && make -j8 > /dev/null 2>&1 ) || { echo "variant_lib: the library itself does not build" >&2; exit 1; }
objs=""
for o in "$csrc"/*.o; do
  keep=1
  for src in "$@"; do [ "$(basename "$o" .o)" = "$(basename "$src" .hip)" ] && keep=0; done
  [ $keep = 1 ] && objs="$objs $o"
done
pids=""
for src in "$@"; do
  ( cd "$csrc" && hipcc $base $flags -c "$src" -o "$tmp/$(basename "$src" .hip).o" ) & pids="$pids $!"
done
for pid in $pids; do wait $pid || { echo "variant_lib: build of $* with '$flags' failed" >&2; exit 1; }; done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$out" $objs "$tmp"/*.o || { echo "variant_lib: link failed" >&2; exit 1; }
