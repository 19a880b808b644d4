#!/bin/bash
# Builds oracle/_ref/libherald_ref.so: oracle/ref_driver.cc (ours) + the reference's own cache-policy
# sources and header-only Unique<T> / MiniLRUCache, compiled from /root/reference where they lie.
# Only possible where /root/reference is mounted; the GPU box uses the prebuilt file.
#
# Also builds oracle/_ref/libref_dnnl.so: the reference's own CPU operators of the hot path,
# src/dnnl_ops/EmbeddingLookup.cpp (cpu_EmbeddingLookup) and src/dnnl_ops/Optimizers.cpp
# (cpu_SGDOptimizerSparseUpdate), compiled UNCHANGED where they lie, with the reference's flags
# (CMakeLists.txt:15,20: -O3 + OpenMP) against the real oneDNN header that PyTorch ships
# (torch/include/dnnl.hpp) -- header only, nothing of oneDNN is linked or called by these two files.
#
# Unbuildable parts of the reference (documented in DESIGN.md): the ps-lite worker/server need
# ZeroMQ + protobuf; laia's schedulers need Boost.  None of those is worked around with stand-ins:
# the PS calls made by src/hetu_cache/src/hetu_client.cc remain undefined symbols of this library and
# are never executed.
set -e
REF=${REF:-/root/reference}
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/_ref"
[ -d "$REF" ] || { echo "no reference at $REF"; exit 0; }
mkdir -p "$OUT"
DNNL="$OUT/libref_dnnl.so"
if [ ! -f "$DNNL" ]; then
  TORCH_INC=$(python3 -c 'import torch,os;print(os.path.join(os.path.dirname(torch.__file__),"include"))')
  [ -f "$TORCH_INC/dnnl.hpp" ] || { echo "no dnnl.hpp under $TORCH_INC"; exit 1; }
  g++ -O3 -fopenmp -shared -fPIC -std=c++17 -w \
    -I"$TORCH_INC" -I"$REF/src/common" -I"$REF/src/dnnl_ops" \
    "$REF/src/dnnl_ops/EmbeddingLookup.cpp" "$REF/src/dnnl_ops/Optimizers.cpp" \
    -o "$DNNL"
  echo "built $DNNL"
fi
TARGET="$OUT/libherald_ref.so"
if [ -f "$TARGET" ] && [ "$TARGET" -nt "$HERE/ref_driver.cc" ]; then
  exit 0
fi
g++ -O2 -std=c++14 -shared -fPIC -w \
  -I"$REF/src/hetu_cache/include" -I"$REF/ps-lite/include" -I"$REF/laia/include" \
  $(python3 -m pybind11 --includes) \
  "$HERE/ref_driver.cc" \
  "$REF/src/hetu_cache/src/lru_cache.cc" "$REF/src/hetu_cache/src/lfu_cache.cc" \
  "$REF/src/hetu_cache/src/lfuopt_cache.cc" "$REF/src/hetu_cache/src/cache.cc" \
  "$REF/src/hetu_cache/src/hetu_client.cc" "$REF/src/hetu_cache/src/embedding.cc" \
  "$REF/ps-lite/src/thread_pool.cc" \
  -o "$TARGET"
echo "built $TARGET"
