#!/bin/sh
# builds tools/bwd_decompositions/libdeepj_bwd_exp.so (gfx950): the pair / two-tile BPTT experiments on top of the
# product's dj_lstm.hip (unity translation unit; dj_current_device comes from dj_common.h)
set -e
here=$(cd "$(dirname "$0")" && pwd)
${HIPCC:-/opt/rocm/bin/hipcc} --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared \
    -o "$here/libdeepj_bwd_exp.so" "$here/lstm_bwd_pair_dual.hip"
echo "$here/libdeepj_bwd_exp.so"
