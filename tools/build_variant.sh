#!/bin/bash
# Dev tool: build a variant of libdeepj_hip.so with extra -D flags (A/B kernel experiments).
#   tools/build_variant.sh NAME [-DFOO=1 ...]   ->  music-generator_amd/lib/libdeepj_hip.NAME.so
# Run with DEEPJ_LIB=<that path>.  Objects of unflagged builds are cached in /tmp/djobj.
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/music-generator_amd/csrc
obj=/tmp/djobj; mkdir -p $obj/$name
pids=()
for f in dj_gemm dj_lstm dj_step dj_elem dj_gen dj_api; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c $src/$f.hip -o $obj/$name/$f.o 2> $obj/$name/$f.log || { grep -m5 error $obj/$name/$f.log; exit 1; } ) &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o $root/music-generator_amd/lib/libdeepj_hip.$name.so $obj/$name/*.o
echo $root/music-generator_amd/lib/libdeepj_hip.$name.so
