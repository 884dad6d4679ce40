// nic.hip — nic::compress (reference nic.hpp:8-101) for a batch of images resident in HBM.
//
// main.cpp plugs nic::compress into frc::collector::collect as the frame compressor
// (main.cpp:113-125; called twice per frame, frc.hpp:118: image and median), so behind the
// real adapter the collect loop is paced by it.  The codec is a byte-oriented RLE whose
// reference implementation is a sequential push/pop state machine; its OUTPUT has a simple
// grammar (oracle/remap_oracle.cpp, orc_nic_compress): maximal runs of >= 3 equal pixels
// become REP tokens, every maximal stretch of shorter runs becomes one LITERAL token.
//
// One workgroup per image, which it takes in LDS-sized segments; inside a segment thread t
// owns a contiguous chunk of pixels and the tokens whose LAST pixel lies in it (details at the
// kernel).  Two block scans per segment, no atomics, output bytes
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

__device__ __noinline__ uint32_t nic_rep_bytes_split(uint32_t len) {
  uint32_t b = 0;
  while (len > 65535u) {  // split: 65535-pixel pieces, never leaving a remainder < 3
    uint32_t piece = 65535u;
    if (len - piece < 3u) piece -= 3u - (len - piece);
    b += 3;
    len -= piece;
  }
  return b + (len <= 6u ? 1u : len <= 255u ? 2u : 3u);
}
__device__ __forceinline__ uint32_t nic_rep_bytes(uint32_t len) {
  return len <= 6u ? 1u : len <= 255u ? 2u : len <= 65535u ? 3u : nic_rep_bytes_split(len);
}
__device__ __noinline__ uint32_t nic_lit_bytes_split(uint32_t n) {
  uint32_t b = 0;
  while (n > 16383u) {  // split: even pieces of 16382 keep the nibble packing aligned
    b += 2 + 8191;
    n -= 16382u;
  }
  return b + (n < 64u ? 1u : 2u) + (n + 1) / 2;
}
__device__ __forceinline__ uint32_t nic_lit_bytes(uint32_t n) {
  return n < 64u ? 1u + ((n + 1) >> 1) : nic_lit_bytes_split(n);
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
template <typename Px>
__device__ __forceinline__ uint8_t* nic_emit_lit(uint8_t* __restrict__ o, Px&& pixel, uint32_t s, uint32_t n) {
  while (n > 0) {
    const uint32_t piece = n > 16383u ? 16382u : n;
    if (piece < 64u) {
      *o++ = (uint8_t)(0x80u | piece);
    } else {
      *o++ = (uint8_t)(0xc0u | (piece >> 8));
      *o++ = (uint8_t)piece;
    }
    for (uint32_t j = 0; j < piece; j += 2)
      *o++ = (uint8_t)(pixel(s + j) << 4 | (j + 1 < piece ? pixel(s + j + 1) : 0u));
    s += piece;
    n -= piece;
  }
  return o;
}

constexpr uint32_t NIC_T = 256;
constexpr uint32_t NIC_CW = 19;                    // dwords per thread and segment (odd: thread t's
                                                   // k-th word sits in bank (19 t + k) mod 64, conflict-free)
constexpr uint32_t NIC_SEG = NIC_T * NIC_CW * 4;   // pixels per segment (19 456; 19 KB of LDS)
constexpr uint32_t NIC_OUT = 1536;                 // bytes of a wave's tokens staged in LDS and written coalesced
                                                   // (a wave-segment makes ~0.5-1.3 KB on the bench streams; 3.6 KB
                                                   // at worst — then, or with tokens reaching far back, it writes
                                                   // straight to global memory); 27 KB per workgroup: five per CU
constexpr uint32_t NIC_WORDS = NIC_T * NIC_CW + 4; // + 4 pixels before, 12 after (stencil reach)

// byte-wise flags (bit 7 of each byte) -> 4 bits
__device__ __forceinline__ uint32_t nic_nib(uint32_t x) { return ((x >> 7) | (x >> 14) | (x >> 21) | (x >> 28)) & 0xfu; }

// The image is taken in segments that fit LDS (coalesced dword loads, one pixel per byte).
// Token boundaries are a LOCAL property of the pixel sequence:
//   eq[i] = px[i] == px[i+1]            t[i] = eq[i] & eq[i+1]  (three equal pixels from i)
//   L[i]  = t[i] | t[i-1] | t[i-2]      (pixel i lies in a run of >= 3)
//   T[i]  = L[i] ? !eq[i] : L[i+1]      (pixel i is the LAST pixel of a token; P - 1 always is)
// so every thread derives the L and T bits of its 76 pixels with byte-parallel ALU ops on
// 19 dwords (no walk, no divergence) and a token's length is the distance to the previous
// T bit — carried across chunks by a max-scan and across segments in LDS.  Sizes -> sum-scan
// -> offsets -> emission, both loops running over the chunk's tokens, not its pixels.
__global__ __launch_bounds__(NIC_T, 6) void nic_compress(NicParams p) {
  __shared__ uint32_t s_px[NIC_WORDS];  // word w <-> pixels seg0 - 4 + 4 w ...
  __shared__ uint8_t s_out[4][NIC_OUT];
  __shared__ int32_t s_max[4];
  __shared__ uint32_t s_sum[4];
  __shared__ uint32_t s_off;  // carried: bytes so far
  __shared__ int32_t s_end;   // carried: last pixel of the last token that ended
  const uint32_t f = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint8_t* px = p.images + (size_t)f * p.pixels;
  const uint32_t P = p.pixels;
  if (tid == 0) {
    s_off = 0;
    s_end = -1;
  }
  bool fits = true;  // once the blob would pass the stride nothing more is written (flagged at the end)

  for (uint32_t seg0 = 0; seg0 < P; seg0 += NIC_SEG) {
    __syncthreads();  // previous segment fully consumed; carried values visible
    // stage [seg0 - 4, seg0 + NIC_SEG + 12): global dword loads at any byte alignment, all in
    // flight before the first LDS store; pixels outside the image become 0x10 / 0x11
    // alternating, equal to nothing next to them (a word that straddles the image's edge is
    // patched byte by byte)
    {
      auto patch = [&](int64_t g) {
        uint32_t v = 0x11101110u;
        for (int b = 0; b < 4; ++b)
          if (g + b >= 0 && g + b < (int64_t)P) v = (v & ~(0xffu << (8 * b))) | ((uint32_t)px[g + b] & 15u) << (8 * b);
        return v;
      };
      uint32_t r[NIC_CW + 1];
      const int32_t g0 = (int32_t)seg0 - 4 + 4 * (int32_t)tid, Pi = (int32_t)P;  // (images < 2^31 pixels)
      if (P >= 8) {
#pragma unroll
        for (int it = 0; it <= (int)NIC_CW; ++it) {
          const int32_t g = g0 + it * (int32_t)(4 * NIC_T);
          const int32_t gc = max(0, min(g, Pi - 4));
          r[it] = (it < (int)NIC_CW || tid < 4) ? *reinterpret_cast<const uint32_t*>(px + gc) & 0x0f0f0f0fu : 0u;
        }
      }
#pragma unroll
      for (int it = 0; it <= (int)NIC_CW; ++it) {
        const int32_t g = g0 + it * (int32_t)(4 * NIC_T);
        if (it < (int)NIC_CW || tid < 4) s_px[tid + it * NIC_T] = (P >= 8 && g >= 0 && g + 4 <= Pi) ? r[it] : patch(g);
      }
    }
    __syncthreads();
    auto at = [&](uint32_t i) {  // i in [seg0 - 4, seg0 + NIC_SEG + 12)
      const int32_t r = (int32_t)(i - seg0);
      return (s_px[(r >> 2) + 1] >> (8 * (r & 3))) & 0xffu;
    };
    const uint32_t off_in = s_off;
    const int32_t end_in = s_end;
    const uint32_t c0 = seg0 + tid * (NIC_CW * 4);  // (may lie beyond P: then no bit is set)
    const uint32_t* d = s_px + tid * NIC_CW + 1;    // d[k] = dword k of the chunk, k in [-1, NIC_CW + 1]

    // ---- L and T bits of the chunk: bit j of the 3-word masks <-> pixel c0 - 4 + j ----
    uint32_t EQ[3] = {0, 0, 0};
#pragma unroll
    for (int k = -1; k <= (int)NIC_CW; ++k) {  // eq flags, 4 per dword (bit 7 of byte b: px[4k+b] == px[4k+b+1])
      const uint32_t a = d[k], x = a ^ __builtin_amdgcn_alignbyte(d[k + 1], a, 1);
      const uint32_t ne = (x + 0x7f7f7f7fu) & 0x80808080u;
      const int j = 4 * (k + 1);
      EQ[j >> 5] |= (nic_nib(ne) ^ 0xfu) << (j & 31);
    }
    // multiword shifts by one / two positions (towards lower = the NEXT pixel moves onto this one)
    auto next1 = [](const uint32_t (&m)[3], uint32_t (&r)[3]) {
      r[0] = __builtin_amdgcn_alignbit(m[1], m[0], 1);
      r[1] = __builtin_amdgcn_alignbit(m[2], m[1], 1);
      r[2] = m[2] >> 1;
    };
    uint32_t Tm[3], Lm[3];
    {
      uint32_t e1[3], t[3], l1[3];
      next1(EQ, e1);
#pragma unroll
      for (int w = 0; w < 3; ++w) t[w] = EQ[w] & e1[w];
      // L = t | t(previous pixel) | t(two back)
      Lm[0] = t[0] | t[0] << 1 | t[0] << 2;
      Lm[1] = t[1] | __builtin_amdgcn_alignbit(t[1], t[0], 31) | __builtin_amdgcn_alignbit(t[1], t[0], 30);
      Lm[2] = t[2] | __builtin_amdgcn_alignbit(t[2], t[1], 31) | __builtin_amdgcn_alignbit(t[2], t[1], 30);
      next1(Lm, l1);
#pragma unroll
      for (int w = 0; w < 3; ++w) Tm[w] = (Lm[w] & ~EQ[w]) | (~Lm[w] & l1[w]);
      // only the chunk's own pixels: j in [4, 4 + 4 * NIC_CW)
      Tm[0] &= ~0xfu;
      Tm[2] &= (1u << (4 + 4 * NIC_CW - 64)) - 1u;
    }
    if (P - 1 >= c0 && P - 1 - c0 < NIC_CW * 4) Tm[(P - 1 - c0 + 4) >> 5] |= 1u << ((P - 1 - c0 + 4) & 31u);
    const int32_t jbase = (int32_t)c0 - 4;  // pixel of bit 0

    // ---- previous token end before the chunk: max-scan of the chunks' last T bits ----
    int32_t last_end = -1;
#pragma unroll
    for (int w = 0; w < 3; ++w)
      if (Tm[w]) last_end = jbase + 32 * w + 31 - __builtin_clz(Tm[w]);
    int32_t mx = last_end;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int32_t t = __shfl_up(mx, o);
      if (lane >= (uint32_t)o) mx = max(mx, t);
    }
    if (lane == 63) s_max[wave] = mx;
    __syncthreads();
    int32_t prev0 = end_in;
    for (uint32_t w = 0; w < wave; ++w) prev0 = max(prev0, s_max[w]);
    {
      const int32_t t = __shfl_up(mx, 1);
      if (lane > 0) prev0 = max(prev0, t);
    }

    // ---- sizes of the tokens that end in the chunk ----
    uint32_t bytes = 0;
    {
      int32_t prev = prev0;
#pragma unroll
      for (int w = 0; w < 3; ++w) {
        uint32_t m = Tm[w];
        while (m) {
          const uint32_t b = (uint32_t)__builtin_ctz(m);
          m &= m - 1;
          const int32_t e = jbase + 32 * w + (int32_t)b;
          const uint32_t len = (uint32_t)(e - prev);
          bytes += ((Lm[w] >> b) & 1u) ? nic_rep_bytes(len) : nic_lit_bytes(len);
          prev = e;
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
    uint32_t off = off_in + sum - bytes;
    for (uint32_t w = 0; w < wave; ++w) off += s_sum[w];
    if (off + bytes > p.stride) fits = false;

    // ---- emission: into the wave's LDS slice, then one coalesced copy per wave ----
    // (a wave whose tokens reach far back — a huge run cut into many pieces, a literal that
    // began chunks ago — may exceed the slice: it writes straight to global memory instead)
    uint32_t wave_off = off_in;
    for (uint32_t w = 0; w < wave; ++w) wave_off += s_sum[w];
    const uint32_t wave_bytes = s_sum[wave];
    const bool staged = wave_bytes <= NIC_OUT;
    const bool wave_fits = wave_off + wave_bytes <= p.stride;
    if (bytes && (staged || fits)) {
      uint8_t* o = staged ? s_out[wave] + (off - wave_off) : p.out + (size_t)f * p.stride + off;
      // literal payload pixels: from the staged segment, or (a literal that began in an
      // earlier segment) from the image in global memory
      auto pixel = [&](uint32_t i) { return i + 4 >= seg0 ? at(i) : ((uint32_t)px[i] & 15u); };
      int32_t prev = prev0;
#pragma unroll
      for (int w = 0; w < 3; ++w) {
        uint32_t m = Tm[w];
        while (m) {
          const uint32_t b = (uint32_t)__builtin_ctz(m);
          m &= m - 1;
          const int32_t e = jbase + 32 * w + (int32_t)b;
          const uint32_t len = (uint32_t)(e - prev);
          if ((Lm[w] >> b) & 1u)
            o = nic_emit_rep(o, at((uint32_t)e), len);
          else
            o = nic_emit_lit(o, pixel, (uint32_t)(prev + 1), len);
          prev = e;
        }
      }
    }
    if (staged && wave_fits) {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      __builtin_amdgcn_wave_barrier();
      uint8_t* dst = p.out + (size_t)f * p.stride + wave_off;
      for (uint32_t j = lane; j < wave_bytes; j += 64) dst[j] = s_out[wave][j];
    }
    if (tid == NIC_T - 1) {  // carry to the next segment
      s_off = off + bytes;
      s_end = max(prev0, last_end);
    }
  }
  __syncthreads();
  if (tid == 0) {
    p.sizes[f] = s_off;
    if (s_off > p.stride) atomicOr(&p.status[2], 1u);
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

// nic::decompress (nic.hpp:103-164): one wave per image.  Token lengths chain through the
// stream, but everything else is done side by side.  Per 64-byte chunk of the stream (read from
// an LDS window): every lane parses the token that WOULD start at its byte; six rounds of
// pointer doubling over those speculative "next token" links mark the real starts and give the
// chunk's exit; wave scans of the real tokens' pixel and piece counts place them; the output is
// cut into 16-pixel pieces handed to the lanes (each finds its token by a 6-step search over the
// piece counts and writes its piece with one unaligned 16-byte store, or 8+4+2+1 for a token's
// tail); tokens over 1024 pixels are written by the whole wave.  History (10 000 sparse 320x240
// images, ~4 100 tokens each): one token per trip with uniform loads 5.5 ms; lane-parallel parse
// with a scalar walk and a per-token wave loop for runs over 8 pixels 9.7 ms.
struct NicDecParams {
  const uint8_t* packed;
  const uint64_t* offsets;
  const uint32_t* sizes;
  uint8_t* images;  // n x pixels
  uint32_t* status;
  uint32_t n, pixels;
};
constexpr uint32_t NIC_WIN = 4096;  // bytes of stream kept in LDS
constexpr uint32_t NIC_BIG = 1024;  // tokens of more pixels are written by the whole wave
constexpr uint32_t NIC_AHEAD = 64 + 2 + NIC_BIG / 2 + 8;  // window bytes a chunk may touch from its first byte

// All lanes fill `len` bytes at `dst` with `v`: bytes up to 16-byte alignment, uint4 body, bytes.
__device__ __forceinline__ void nic_wave_fill(uint8_t* dst, uint32_t len, uint32_t v, uint32_t lane) {
  const uint32_t head = min(len, (uint32_t)(-(intptr_t)reinterpret_cast<uintptr_t>(dst)) & 15u);
  if (lane < head) dst[lane] = (uint8_t)v;
  const uint32_t nvec = (len - head) >> 4;
  const uint32_t w = v * 0x01010101u;
  uint4* body = reinterpret_cast<uint4*>(dst + head);
  for (uint32_t j = lane; j < nvec; j += 64) body[j] = make_uint4(w, w, w, w);
  const uint32_t done = head + 16 * nvec;
  if (lane < len - done) dst[done + lane] = (uint8_t)v;
}

__global__ __launch_bounds__(64) void nic_decompress(NicDecParams p) {
  __shared__ uint32_t win[NIC_WIN / 4 + 2];
  __shared__ uint32_t reach[64];
  const uint32_t f = blockIdx.x, lane = threadIdx.x;
  const uint8_t* src = p.packed + p.offsets[f];
  const uint32_t size = p.sizes[f];
  uint8_t* out = p.images + (size_t)f * p.pixels;
  const int32_t mis = (int32_t)(reinterpret_cast<uintptr_t>(src) & 3u);  // window words are aligned dwords of memory
  int32_t wbase = INT32_MIN;  // stream offset of win[0]; -3..-1 when the stream starts inside its first dword
  uint32_t i = 0, o = 0;
  bool bad = size >= 0x7fff0000u;
  auto lds_sync = [] {  // one wave: LDS is in order, only the compiler has to be told
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
  };
  auto refill = [&](uint32_t k) {  // window from byte k on (uniform)
    wbase = ((int32_t)k + mis) / 4 * 4 - mis;
    lds_sync();
    const uint8_t* g = src + wbase;  // 4-byte aligned
    const int32_t avail = (int32_t)size - wbase;  // bytes of the stream from win[0] on
    for (uint32_t w = lane; w < NIC_WIN / 4 && (int32_t)(4 * w) < avail; w += 64) {
      uint32_t v;
      if ((int32_t)(4 * w + 4) <= avail) {
        v = *reinterpret_cast<const uint32_t*>(g + 4 * w);
      } else {  // the stream ends inside this dword: nothing past its last byte is touched
        v = 0;
        for (int32_t b = 0; (int32_t)(4 * w) + b < avail; ++b) v |= (uint32_t)g[4 * w + b] << (8 * b);
      }
      win[w] = v;
    }
    lds_sync();
  };
  auto dword_at = [&](uint32_t k) -> uint32_t {  // stream bytes k..k+3 out of the window
    const uint32_t r = min((uint32_t)((int32_t)k - wbase), NIC_WIN);
    return __builtin_amdgcn_alignbyte(win[(r >> 2) + 1], win[r >> 2], r & 3u);
  };
  while (!bad && i < size) {
    if ((int32_t)i < wbase || (int64_t)i + NIC_AHEAD > (int64_t)wbase + NIC_WIN) refill(i);
    // ---- the token that would start at byte i + lane ----
    const uint32_t k = i + lane;
    const uint32_t hv = dword_at(k);
    const uint32_t t = hv & 0xffu;
    const bool lit = (t & 0x80u) != 0;
    uint32_t tl, px;
    bool ok = k < size;
    if (!lit) {
      const uint32_t nb = (t & 0x40u) ? (t >> 4) & 3u : 0u;
      tl = 1 + nb;
      px = (t & 0x40u) ? (hv >> 8) & (nb ? 0xffffffu >> (8 * (3 - nb)) : 0u) : (t >> 4) + 3;
      ok = ok && k + nb < size;
    } else {
      px = t & 0x3fu;
      tl = 1;
      if (t & 0x40u) {
        px = (px << 8) + ((hv >> 8) & 0xffu);
        tl = 2;
      }
      tl += (px + 1) >> 1;
      ok = ok && (uint64_t)k + tl <= size;
    }
    // ---- the real token starts: pointer doubling over "next token" from lane 0 ----
    // link < 64: a lane of this chunk; otherwise absorbing: the chunk's exit offset, 0x10000 = a
    // broken token, 0x20000 + lane = the end of the stream
    uint32_t link = ok ? lane + tl : (k < size ? 0x10000u : 0x20000u + lane);
    reach[lane] = lane == 0;
#pragma unroll
    for (int round = 0; round < 6; ++round) {
      lds_sync();
      const uint32_t r = reach[lane];
      if (r && link < 64) reach[link] = 1;
      const uint32_t far = (uint32_t)__shfl((int)link, (int)(link & 63u));
      if (link < 64) link = far;
    }
    lds_sync();
    const bool mine = reach[lane] != 0 && k < size;
    const uint32_t exit_code = (uint32_t)__builtin_amdgcn_readlane((int)link, 0);
    if (exit_code == 0x10000u) {
      bad = true;
      break;
    }
    const uint32_t pos = exit_code >= 0x20000u ? exit_code - 0x20000u : exit_code;
    // ---- placement: scans of the real tokens' pixels and 16-pixel pieces ----
    const bool big = mine && px > NIC_BIG;
    uint32_t ipx = mine ? px : 0u;
    uint32_t ipc = (mine && !big) ? (px + 15) >> 4 : 0u;
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1) {
      const uint32_t u = (uint32_t)__shfl_up((int)ipx, d), v = (uint32_t)__shfl_up((int)ipc, d);
      if (lane >= d) {
        ipx += u;
        ipc += v;
      }
    }
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)ipx, 63);
    const uint32_t pieces = (uint32_t)__builtin_amdgcn_readlane((int)ipc, 63);
    if (total > p.pixels - o) {
      bad = true;
      break;
    }
    const uint32_t epx = ipx - (mine ? px : 0u);  // pixel offset of my token in the chunk
    // ---- pieces: lane w writes piece w ----
    for (uint32_t base = 0; base < pieces; base += 64) {
      const uint32_t w = base + lane;
      uint32_t s = 0;  // the first lane whose inclusive piece count exceeds w: the piece's token
#pragma unroll
      for (uint32_t step = 32; step > 0; step >>= 1) {
        const uint32_t v = (uint32_t)__shfl((int)ipc, (int)(s + step - 1));
        if (v <= w) s += step;
      }
      s = min(s, 63u);
      const uint32_t spx = (uint32_t)__shfl((int)px, (int)s);
      const uint32_t shv = (uint32_t)__shfl((int)hv, (int)s);
      const uint32_t sepx = (uint32_t)__shfl((int)epx, (int)s);
      const uint32_t sipc = (uint32_t)__shfl((int)ipc, (int)s);
      if (w < pieces) {
        const uint32_t pi = w - (sipc - ((spx + 15) >> 4));
        uint32_t cnt = min(16u, spx - 16 * pi);
        uint8_t* d = out + o + sepx + 16 * pi;
        uint32_t d0, d1, d2, d3;
        if ((shv & 0x80u) == 0) {
          d0 = d1 = d2 = d3 = (shv & 15u) * 0x01010101u;
        } else {  // 8 payload bytes, high nibble first
          const uint32_t at = i + s + ((shv & 0x40u) ? 2u : 1u) + 8 * pi;
          const uint32_t x0 = dword_at(at), x1 = dword_at(at + 4);
          const uint32_t e0 = (x0 >> 4) & 0x0f0f0f0fu, o0 = x0 & 0x0f0f0f0fu;
          const uint32_t e1 = (x1 >> 4) & 0x0f0f0f0fu, o1 = x1 & 0x0f0f0f0fu;
          d0 = perm(o0, e0, 0x05010400u);
          d1 = perm(o0, e0, 0x07030602u);
          d2 = perm(o1, e1, 0x05010400u);
          d3 = perm(o1, e1, 0x07030602u);
        }
        if (cnt == 16) {
          const uint4 v = make_uint4(d0, d1, d2, d3);
          __builtin_memcpy(d, &v, 16);  // unaligned global_store_dwordx4
        } else {
          if (cnt & 8u) {
            const uint2 v = make_uint2(d0, d1);
            __builtin_memcpy(d, &v, 8);
            d0 = d2;
            d1 = d3;
            d += 8;
          }
          if (cnt & 4u) {
            __builtin_memcpy(d, &d0, 4);
            d0 = d1;
            d += 4;
          }
          if (cnt & 2u) {
            const uint16_t v = (uint16_t)d0;
            __builtin_memcpy(d, &v, 2);
            d0 >>= 16;
            d += 2;
          }
          if (cnt & 1u) *d = (uint8_t)d0;
        }
      }
    }
    // ---- tokens over NIC_BIG pixels: the whole wave writes each ----
    uint64_t bigs = __ballot(big);
    while (bigs) {
      const int sl = __builtin_ctzll(bigs);
      bigs &= bigs - 1;
      const uint32_t lt = (uint32_t)__builtin_amdgcn_readlane((int)t, sl);
      const uint32_t ln = (uint32_t)__builtin_amdgcn_readlane((int)px, sl);
      uint8_t* d = out + o + (uint32_t)__builtin_amdgcn_readlane((int)epx, sl);
      if ((lt & 0x80u) == 0) {
        nic_wave_fill(d, ln, lt & 15u, lane);
      } else {
        const uint8_t* pay = src + i + (uint32_t)sl + ((lt & 0x40u) ? 2u : 1u);
        for (uint32_t j = lane; 2 * j < ln; j += 64) {  // a payload byte = two pixels per lane
          const uint32_t b = pay[j];
          d[2 * j] = (uint8_t)(b >> 4);
          if (2 * j + 1 < ln) d[2 * j + 1] = (uint8_t)(b & 15u);
        }
      }
    }
    o += total;
    i += pos;
  }
  if ((bad || o != p.pixels) && lane == 0) atomicOr(&p.status[2], 2u);  // malformed stream / wrong pixel count
}

}  // namespace rmx
