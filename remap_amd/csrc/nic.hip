// nic.hip — nic::compress (reference nic.hpp:8-101) for a batch of images resident in HBM.
//
// main.cpp plugs nic::compress into frc::collector::collect as the frame compressor
// (main.cpp:113-125; called twice per frame, frc.hpp:118: image and median), so behind the
// real adapter the collect loop is paced by it.  The codec is a byte-oriented RLE whose
// reference implementation is a sequential push/pop state machine; its OUTPUT has a simple
// grammar (oracle/remap_oracle.cpp, orc_nic_compress): maximal runs of >= 3 equal pixels
// become REP tokens, every maximal stretch of shorter runs becomes one LITERAL token.
//
// One workgroup per image; thread t owns a contiguous chunk of pixels.  Tokens are
// attributed to the thread that owns their LAST pixel, which can size them from two carried
// facts: how long the run reaching into its chunk already is (segmented scan over chunk
// summaries) and where the last completed long run ended (max-scan).  A sum-scan of the
// per-chunk byte counts then gives every token its output offset — three block scans and
// three sequential walks over 1/256 of the image per thread, no atomics, output bytes
// identical to the reference's for every input inside its domain (runs <= 65535 pixels,
// literals <= 16383); beyond it, where the reference overflows its uint16 counters, runs and
// literals are split into several well-formed tokens (same rule as the oracle).
#include "common.h"

namespace rmx {

struct NicParams {
  const uint8_t* images;  // n x pixels
  uint8_t* out;           // n x stride
  uint32_t* sizes;        // n compressed sizes (bytes)
  uint32_t* status;
  uint32_t n, pixels, stride;
};

__device__ __forceinline__ uint32_t nic_rep_bytes(uint32_t len) {
  uint32_t b = 0;
  while (len > 65535u) {  // split: 65535-pixel pieces, never leaving a remainder < 3
    uint32_t piece = 65535u;
    if (len - piece < 3u) piece -= 3u - (len - piece);
    b += 3;
    len -= piece;
  }
  return b + (len <= 6u ? 1u : len <= 255u ? 2u : 3u);
}
__device__ __forceinline__ uint32_t nic_lit_bytes(uint32_t n) {
  uint32_t b = 0;
  while (n > 16383u) {  // split: even pieces of 16382 keep the nibble packing aligned
    b += 2 + 8191;
    n -= 16382u;
  }
  return b + (n < 64u ? 1u : 2u) + (n + 1) / 2;
}
__device__ inline uint8_t* nic_emit_rep(uint8_t* o, uint32_t v, uint32_t len) {
  while (len > 0) {
    uint32_t piece = len;
    if (len > 65535u) {
      piece = 65535u;
      if (len - piece < 3u) piece -= 3u - (len - piece);
    }
    if (piece <= 6u) {
      *o++ = (uint8_t)(((piece - 3u) << 4) | v);
    } else if (piece <= 255u) {
      *o++ = (uint8_t)(0x50u | v);
      *o++ = (uint8_t)piece;
    } else {
      *o++ = (uint8_t)(0x60u | v);
      *o++ = (uint8_t)piece;
      *o++ = (uint8_t)(piece >> 8);
    }
    len -= piece;
  }
  return o;
}
__device__ inline uint8_t* nic_emit_lit(uint8_t* o, const uint8_t* px, uint32_t n) {
  while (n > 0) {
    const uint32_t piece = n > 16383u ? 16382u : n;
    if (piece < 64u) {
      *o++ = (uint8_t)(0x80u | piece);
    } else {
      *o++ = (uint8_t)(0xc0u | (piece >> 8));
      *o++ = (uint8_t)piece;
    }
    for (uint32_t j = 0; j < piece; j += 2)
      *o++ = (uint8_t)((px[j] & 15u) << 4 | (j + 1 < piece ? (px[j + 1] & 15u) : 0u));
    px += piece;
    n -= piece;
  }
  return o;
}

// 256-thread inclusive scans through wave shuffles + four wave totals
struct NicSeg {
  uint32_t reset, x;  // x = run length reaching the chunk's last pixel
};
__device__ __forceinline__ NicSeg nic_join(const NicSeg& a, const NicSeg& b) {
  return NicSeg{a.reset | b.reset, b.reset ? b.x : a.x + b.x};
}

constexpr uint32_t NIC_T = 256;

__global__ __launch_bounds__(NIC_T) void nic_compress(NicParams p) {
  __shared__ NicSeg s_seg[4];
  __shared__ int32_t s_max[4];
  __shared__ uint32_t s_sum[4];
  const uint32_t f = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint8_t* px = p.images + (size_t)f * p.pixels;
  const uint32_t P = p.pixels;
  const uint32_t cp = (P + NIC_T - 1) / NIC_T;
  const uint32_t c0 = min(P, tid * cp), c1 = min(P, c0 + cp);
  const bool has = c1 > c0;
  auto at = [&](uint32_t i) { return (uint32_t)px[i] & 15u; };

  // ---- walk 1: chunk summary (is it one run, and the run reaching its last pixel) ----
  uint32_t tail = 0;
  bool uniform = true;
  if (has) {
    uint32_t v = at(c0);
    tail = 1;
    for (uint32_t i = c0 + 1; i < c1; ++i) {
      const uint32_t w = at(i);
      if (w == v) {
        ++tail;
      } else {
        v = w;
        tail = 1;
        uniform = false;
      }
    }
  }
  // run length at the chunk end = own length + the predecessor's, iff the chunk is one run
  // that continues the previous pixel; otherwise the in-chunk tail (a reset)
  const bool cont = has && c0 > 0 && at(c0 - 1) == at(c0);
  NicSeg inc{(uint32_t)!(uniform && cont), tail};
  if (!has) inc = NicSeg{0u, 0u};
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    NicSeg t{(uint32_t)__shfl_up((int)inc.reset, o), (uint32_t)__shfl_up((int)inc.x, o)};
    if (lane >= (uint32_t)o) inc = nic_join(t, inc);
  }
  if (lane == 63) s_seg[wave] = inc;
  __syncthreads();
  NicSeg pre{0u, 0u};
  for (uint32_t w = 0; w < wave; ++w) pre = nic_join(pre, s_seg[w]);
  NicSeg before{(uint32_t)__shfl_up((int)inc.reset, 1), (uint32_t)__shfl_up((int)inc.x, 1)};
  if (lane == 0) before = NicSeg{0u, 0u};
  const NicSeg left = nic_join(pre, before);  // run reaching pixel c0 - 1
  const uint32_t rl0 = cont ? left.x : 0u;    // pixels of c0's run that lie before the chunk

  // a run is visited at its LAST pixel e (e == P - 1 or px[e + 1] differs)
  // ---- walk 2: last long-run end inside the chunk ----
  int32_t last_long = -1;
  if (has) {
    uint32_t v = at(c0), len = rl0 + 1;
    for (uint32_t i = c0; i < c1; ++i) {
      const bool end = i + 1 == P || at(i + 1) != v;
      if (end) {
        if (len >= 3) last_long = (int32_t)i;
        if (i + 1 < P) v = at(i + 1);
        len = 1;
      } else {
        ++len;
      }
    }
  }
  int32_t mx = last_long;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int32_t t = __shfl_up(mx, o);
    if (lane >= (uint32_t)o) mx = max(mx, t);
  }
  if (lane == 63) s_max[wave] = mx;
  __syncthreads();
  int32_t carried = -1;
  for (uint32_t w = 0; w < wave; ++w) carried = max(carried, s_max[w]);
  {
    const int32_t t = __shfl_up(mx, 1);
    if (lane > 0) carried = max(carried, t);
  }
  // carried = last pixel of the last long run that ended before this chunk (-1: none)

  // ---- walk 3 / 4: size, then emit, the tokens that END in this chunk ----
  // A literal ends at e if e closes a short run and (e == P - 1 or a long run starts at e + 1).
  auto long_starts = [&](uint32_t i) {  // run starting at i has >= 3 pixels
    return i + 2 < P && at(i + 1) == at(i) && at(i + 2) == at(i);
  };
  uint32_t bytes = 0;
  if (has) {
    uint32_t v = at(c0), len = rl0 + 1;
    int32_t ll = carried;
    for (uint32_t i = c0; i < c1; ++i) {
      const bool end = i + 1 == P || at(i + 1) != v;
      if (end) {
        if (len >= 3) {
          bytes += nic_rep_bytes(len);
          ll = (int32_t)i;
        } else if (i + 1 == P || long_starts(i + 1)) {
          bytes += nic_lit_bytes(i - (uint32_t)(ll + 1) + 1);
        }
        if (i + 1 < P) v = at(i + 1);
        len = 1;
      } else {
        ++len;
      }
    }
  }
  uint32_t sum = bytes;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = (uint32_t)__shfl_up((int)sum, o);
    if (lane >= (uint32_t)o) sum += t;
  }
  if (lane == 63) s_sum[wave] = sum;
  __syncthreads();
  uint32_t off = sum - bytes;
  for (uint32_t w = 0; w < wave; ++w) off += s_sum[w];
  if (tid == NIC_T - 1) {
    const uint32_t total = off + bytes;
    p.sizes[f] = total;
    if (total > p.stride) atomicOr(&p.status[2], 1u);
  }
  if (has && off + bytes <= p.stride) {  // (a frame that does not fit is flagged, nothing of this chunk is written)
    uint8_t* o = p.out + (size_t)f * p.stride + off;
    uint32_t v = at(c0), len = rl0 + 1;
    int32_t ll = carried;
    for (uint32_t i = c0; i < c1; ++i) {
      const bool end = i + 1 == P || at(i + 1) != v;
      if (end) {
        if (len >= 3) {
          o = nic_emit_rep(o, v, len);
          ll = (int32_t)i;
        } else if (i + 1 == P || long_starts(i + 1)) {
          const uint32_t s = (uint32_t)(ll + 1);
          o = nic_emit_lit(o, px + s, i - s + 1);
        }
        if (i + 1 < P) v = at(i + 1);
        len = 1;
      } else {
        ++len;
      }
    }
  }
}

// Contiguous packing: blob f (sizes[f] bytes at out + f * stride) -> packed + offsets[f].
struct NicPackParams {
  const uint8_t* blobs;
  const uint32_t* sizes;
  const uint64_t* offsets;
  uint8_t* packed;
  uint32_t n, stride;
};
__global__ __launch_bounds__(256) void nic_pack(NicPackParams p) {
  const uint32_t f = blockIdx.x;
  const uint8_t* src = p.blobs + (size_t)f * p.stride;
  uint8_t* dst = p.packed + p.offsets[f];
  const uint32_t n = min(p.sizes[f], p.stride);
  for (uint32_t i = threadIdx.x; i < n; i += 256) dst[i] = src[i];
}

}  // namespace rmx
