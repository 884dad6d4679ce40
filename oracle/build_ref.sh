#!/usr/bin/env bash
# build_ref.sh — compile the REFERENCE's own hot-path headers into oracle/_ref/*.so so the
# CPU restatement (remap_oracle.cpp) can be validated against them and golden vectors can be
# generated (oracle/make_golden.py).  Runs only where /root/reference exists (this
# container); the GPU box only ever sees the prebuilt .so files (oracle/_ref/ is
# git-ignored but not gpurun-ignored).  No reference source is written into the repo.
#
# Tier 1  libref_strict.so : cpl/kpr/kpm/fgm headers, UNMODIFIED, compiled in place.
# Tier 2  libref_full.so   : + kpe/frc/fgs.  kpe.hpp and all.hpp are MSVC-only as written;
#         they are compiled from a temporary copy outside the repo with four one-line
#         spelling patches that do not change behaviour:
#           kpe.hpp:10   <intrin.h>                      -> <immintrin.h>
#           kpe.hpp:55   _mm256_castsi128_si256({})      -> (_mm_setzero_si128())
#           kpe.hpp:381-394  three dead members (never called) deleted
#           all.hpp:158  `= {n}` through an explicit ctor -> `= memory_pool{n}`
# Tier 2  libref_fdf.so    : + fdf/fde/cte (foreground filter), see the end of this script.
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
ref="${REMAP_REFERENCE:-/root/reference}/src"
out="$here/_ref"
if [ ! -d "$ref" ]; then
  echo "build_ref: $ref absent — keeping prebuilt oracle/_ref (if any)"; exit 0
fi
mkdir -p "$out"
CXX="${CXX:-g++}"
FLAGS="-std=c++20 -O2 -mavx2 -fpermissive -w -shared -fPIC"

# ---- tier 1: in place, unmodified -----------------------------------------------------------
$CXX $FLAGS -I"$ref" "$here/ref_harness_strict.cpp" -o "$out/libref_strict.so"
echo "build_ref: built $out/libref_strict.so (tier 1, unmodified headers)"

# ---- tier 2: temporary patched copy of kpe.hpp / all.hpp -------------------------------------
tmp="$(mktemp -d /tmp/remap_ref.XXXXXX)"
trap 'rm -rf "$tmp"' EXIT
cp "$ref"/*.hpp "$tmp/"
sed -i '10s/#include <intrin.h>/#include <immintrin.h>/' "$tmp/kpe.hpp"
sed -i '55s/_mm256_castsi128_si256({})/_mm256_castsi128_si256(_mm_setzero_si128())/' "$tmp/kpe.hpp"
sed -i '381,394d' "$tmp/kpe.hpp"
sed -i '158s/\*current_ = {previous_->total_used() << 1};/*current_ = memory_pool{previous_->total_used() << 1};/' "$tmp/all.hpp"
# the four patches must have applied, and nothing else may differ
n=$(diff <(cat "$ref/kpe.hpp" "$ref/all.hpp") <(cat "$tmp/kpe.hpp" "$tmp/all.hpp") | grep -c '^[<>]' || true)
if [ "$n" != "20" ]; then echo "build_ref: unexpected patch size ($n changed lines)"; exit 1; fi
$CXX $FLAGS -I"$tmp" "$here/ref_harness_full.cpp" -o "$out/libref_full.so"
echo "build_ref: built $out/libref_full.so (tier 2, patched temporary copy, removed)"

# ---- tier 2, foreground filter (fdf/fde/cte): one more spelling patch, AVX-512VL/BW flags ------
#   fde.hpp:10   <intrin.h>  -> <immintrin.h>
# Built only where the host CPU can run it; loaded only by tests / make_golden on this machine.
if grep -q avx512bw /proc/cpuinfo && grep -q avx512vl /proc/cpuinfo; then
  sed -i '10s/#include <intrin.h>/#include <immintrin.h>/' "$tmp/fde.hpp"
  n=$(diff "$ref/fde.hpp" "$tmp/fde.hpp" | grep -c '^[<>]' || true)
  if [ "$n" != "2" ]; then echo "build_ref: unexpected fde.hpp patch size ($n changed lines)"; exit 1; fi
  $CXX $FLAGS -mavx512vl -mavx512bw -Wl,-Bsymbolic -I"$tmp" "$here/ref_harness_fdf.cpp" -o "$out/libref_fdf.so"
  echo "build_ref: built $out/libref_fdf.so (tier 2, foreground filter)"
else
  echo "build_ref: no AVX-512VL/BW on this host - libref_fdf.so not built"
fi
