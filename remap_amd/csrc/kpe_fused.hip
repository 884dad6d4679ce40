// kpe_fused.hip — keypoint extraction in ONE streaming kernel (gfx950).
//
// kpe.hip's kpe_main + kpe_finish pair reads the frames twice (the second time through
// latency-bound gathers that touch nearly every 128-B line again) and is bound by its memory
// access pattern, not by arithmetic (tools/micro/tile_bw.hip).  Here a
// workgroup owns a TILE = FPB frames x R output rows:
//
//   load     rows y0-2 .. y0+R+1 of every frame of the tile are one contiguous piece of HBM:
//            read once with 16 B per lane, all loads of a lane in flight together, mapped to
//            KEYS (rank << 4 | native: one LUT per 4 pixels, common.h) and kept in LDS;
//   3x3      a lane owns a dword column (4 pixels) and walks down the tile's rows with three
//            rows of sorted triples in registers — the rank network of kpe.hip (7 ops per
//            pixel) — but its operands are LDS byte reads: no unpacking, no cross-lane moves,
//            no halo lanes; the median row goes straight to HBM, the "centre differs" flags of
//            the whole walk stay in four registers;
//   5x5      the flagged pixels of the tile (1-8 % on the synthetic streams) are numbered by a
//            block scan, queued in LDS and handed out densely, one per lane: the 5x5 window
//            comes out of the key tile (ten LDS reads, no HBM, no LUT), 12th-largest-of-25 by a
//            4-step binary search on the rank with SWAR byte compares, weight, code bytes
//            (kpe.hpp:316-321,342-379), append.
//
// Every frame byte is read from HBM once (+ 4 halo rows per R, served by the XCD's L2: the
// block index is remapped so that neighbouring strips run on one XCD), every median byte
// written once; there is no candidate plane and no second launch.
// Used for frame widths that are a multiple of 4 and whose tile fits the LDS; kpe.hip's pair
// remains for the other shapes.
#include "common.h"

namespace rmx {

struct KpeFusedParams {
  const uint8_t* frames;
  uint8_t* median;
  RawDesc* raw;
  uint32_t* kcount;
  uint32_t* status;  // [0] overflow flag
  uint32_t n, W, H, Wd, kmax;
  uint32_t fpb;     // frames per workgroup (fpb * Wd <= blockDim.x)
  uint32_t strips;  // ceil(H / R)
  uint32_t pitch;   // bytes between the key tiles of two frames of a workgroup (multiple of 16)
};

constexpr uint32_t KF_QCAP = 1024;  // candidate queue entries per pass

// LDS bytes of a launch: lead pad, key tiles, tail pad, queue, scan scratch
__host__ __device__ inline uint32_t kf_lds_bytes(uint32_t fpb, uint32_t pitch) { return 16 + fpb * pitch + 16 + KF_QCAP * 4 + 128; }

__device__ __forceinline__ uint32_t lshl_or(uint32_t a, uint32_t sh, uint32_t b) {
  uint32_t d;
  asm("v_lshl_or_b32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(sh), "v"(b));
  return d;
}

struct Tri4 {
  uint32_t lo[4], md[4], hi[4];
};

// sorted horizontal triples of the four pixels of a lane's dword in one tile row; q = LDS
// address of the byte left of the dword
__device__ __forceinline__ void tri_row(const uint8_t* q, Tri4& T) {
  const uint32_t kl = q[0], k0 = q[1], k1 = q[2], k2 = q[3], k3 = q[4], kr = q[5];
  T.lo[0] = umin3(kl, k0, k1), T.md[0] = umed3(kl, k0, k1), T.hi[0] = umax3(kl, k0, k1);
  T.lo[1] = umin3(k0, k1, k2), T.md[1] = umed3(k0, k1, k2), T.hi[1] = umax3(k0, k1, k2);
  T.lo[2] = umin3(k1, k2, k3), T.md[2] = umed3(k1, k2, k3), T.hi[2] = umax3(k1, k2, k3);
  T.lo[3] = umin3(k2, k3, kr), T.md[3] = umed3(k2, k3, kr), T.hi[3] = umax3(k2, k3, kr);
}

// 4th largest of the 3x3 window (median_pixel(H3, 4), kpe.hpp:313,326-340) of four pixels,
// packed keys; the network of kpe.hip (tests/test_host_logic.py verifies it exhaustively)
__device__ __forceinline__ uint32_t merge4(const Tri4& A, const Tri4& B, const Tri4& C) {
  uint32_t r[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t a = umax3(A.lo[i], B.lo[i], C.md[i]);
    const uint32_t b = umed3(A.md[i], B.md[i], C.hi[i]);
    const uint32_t c = umed3(A.hi[i], B.hi[i], C.lo[i]);
    r[i] = umed3(a, b, c);
  }
  return lshl_or(lshl_or(r[3], 8u, r[2]), 16u, lshl_or(r[1], 8u, r[0]));
}

// R = output rows of a tile (multiple of 4), V16 = rows are 16-byte multiples (W % 16 == 0):
// the tile is loaded as one linear piece per frame with 16 B per lane; otherwise every lane
// loads the dwords of its own column.
template <int R, bool V16>
__global__ __launch_bounds__(1024) void kpe_fused(KpeFusedParams p) {
  extern __shared__ __attribute__((aligned(16))) uint8_t kf_smem[];
  constexpr int TR = R + 4;  // tile rows: image rows y0-2 .. y0+R+1
  static_assert(R % 4 == 0 && R >= 4 && R <= 24, "R");
  uint8_t* const tile = kf_smem + 16;
  uint32_t* const q1 = reinterpret_cast<uint32_t*>(kf_smem + 32 + p.fpb * p.pitch);
  uint32_t* const ctl = q1 + KF_QCAP;  // [0..15] wave totals of the scan
  const uint32_t NT = blockDim.x, tid = threadIdx.x;
  const uint32_t lane = tid & 63u, wv = tid >> 6, nwv = NT >> 6;

  // XCD-aware block remap (bijective): the blocks of one XCD (bid % 8) walk a contiguous range
  // of tiles, strip after strip, so the halo rows two strips share are found in that XCD's L2.
  const uint32_t nb = gridDim.x, bid = blockIdx.x;
  const uint32_t bq = nb >> 3, br = nb & 7u, xcd = bid & 7u;
  const uint32_t lb = (xcd < br ? xcd * (bq + 1) : br * (bq + 1) + (xcd - br) * bq) + (bid >> 3);
  const uint32_t fg = lb / p.strips, s = lb - fg * p.strips;
  const int H = (int)p.H, W = (int)p.W;
  const int y0 = (int)s * R;
  const int ya = max(0, y0 - 2), yb = min(H, y0 + R + 2);  // image rows held by the tile
  const uint32_t f0 = fg * p.fpb;
  const uint32_t nf = min(p.fpb, p.n - f0);
  const size_t frame_bytes = (size_t)W * H;

  // this lane's dword column
  const uint32_t j = tid / p.Wd, c = tid - j * p.Wd;
  const bool active = j < nf;
  const uint32_t f = f0 + j;

  // ---- load: HBM -> keys -> LDS ------------------------------------------------------------------
  if (V16) {
    constexpr int CH = (TR + 3) / 4;  // 16-byte chunks per lane: fpb * Wd <= NT
    const uint32_t n16 = (uint32_t)((yb - ya) * W) >> 4;
    const uint32_t total = nf * n16;
    const uint32_t toff = (uint32_t)((ya - (y0 - 2)) * W);
    const float inv = 1.0f / (float)n16;
    uint4 v[CH];
    uint32_t dst[CH];
#pragma unroll
    for (int k = 0; k < CH; ++k) {
      const uint32_t q = tid + (uint32_t)k * NT;
      dst[k] = 0xffffffffu;
      if (q < total) {
        uint32_t jj = (uint32_t)(((float)q + 0.5f) * inv);
        jj = jj * n16 > q ? jj - 1 : jj;  // (never taken for the sizes in use; cheap insurance)
        jj = (jj + 1) * n16 <= q ? jj + 1 : jj;
        const uint32_t i = q - jj * n16;
        v[k] = *reinterpret_cast<const uint4*>(p.frames + (size_t)(f0 + jj) * frame_bytes + (size_t)ya * W + (size_t)i * 16);
        dst[k] = jj * p.pitch + toff + i * 16;
      }
    }
#pragma unroll
    for (int k = 0; k < CH; ++k)
      if (dst[k] != 0xffffffffu) {
        uint4 u = v[k];
        u.x = map_key(u.x), u.y = map_key(u.y), u.z = map_key(u.z), u.w = map_key(u.w);
        *reinterpret_cast<uint4*>(tile + dst[k]) = u;
      }
  } else {
    uint32_t w[TR];
    const uint8_t* src = p.frames + (size_t)f * frame_bytes + 4 * c;
#pragma unroll
    for (int tr = 0; tr < TR; ++tr) {
      const int y = y0 - 2 + tr;
      w[tr] = 0;
      if (active && y >= ya && y < yb) w[tr] = *reinterpret_cast<const uint32_t*>(src + (size_t)y * W);
    }
    if (active) {
#pragma unroll
      for (int tr = 0; tr < TR; ++tr) *reinterpret_cast<uint32_t*>(tile + j * p.pitch + (uint32_t)(tr * W) + 4 * c) = map_key(w[tr]);
    }
  }
  __syncthreads();

  // ---- 3x3 pass: medians to HBM, flags into registers -------------------------------------------
  // Cw[q]: rows 4q .. 4q+3 of the tile's output rows, row 4q+k at bit 7-k of pixel i's byte
  uint32_t Cw[R / 4];
#pragma unroll
  for (int q = 0; q < R / 4; ++q) Cw[q] = 0;
#ifdef RMX_KF_NOMAIN  // experiment (tools/): load phase only
  if (false) {
#else
  if (active) {
#endif
    // per-pixel x validity: processed x in [2, W-3] (kpe.hpp:183-184)
    uint32_t maskx = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const uint32_t x = c * 4 + i;
      if (x >= 2 && x + 3 <= p.W) maskx |= 0x0fu << (8 * i);
    }
    const uint8_t* q = tile + j * p.pitch + 4 * c - 1;  // byte left of the lane's dword, tile row 0
    uint8_t* out = p.median + (size_t)f * frame_bytes + (size_t)y0 * W + 4 * c;
    Tri4 T0, T1, T2;
    tri_row(q + W, T0);      // tile row 1 = image row y0 - 1
    tri_row(q + 2 * W, T1);  // tile row 2 = image row y0
    const uint8_t* qn = q + 3 * W;
    uint32_t C = 0;
#define RMX_ROW(A, B, Cn, r)                                                                       \
  {                                                                                                \
    const int y = y0 + (r);                                                                        \
    tri_row(qn, Cn);                                                                               \
    const uint32_t key = *reinterpret_cast<const uint32_t*>(qn - W + 1); /* the centre row */      \
    qn += W;                                                                                       \
    const bool valid = y >= 2 && y < H - 4; /* processed rows: [2, H-5] (kpe.hpp:268) */           \
    const uint32_t pk = merge4(A, B, Cn) & (valid ? maskx : 0u);                                   \
    if (y < H) *reinterpret_cast<uint32_t*>(out + (size_t)(r) * W) = pk;                          \
    const uint32_t t = ((((pk ^ key) & (valid ? maskx : 0u)) + 0x0f0f0f0fu) & 0x10101010u);        \
    C = (C + C) | t;                                                                               \
    if (((r) & 3) == 3) {                                                                          \
      Cw[(r) >> 2] = C;                                                                            \
      C = 0;                                                                                       \
    }                                                                                              \
  }
#pragma unroll
    for (int r = 0; r < R; r += 3) {
      RMX_ROW(T0, T1, T2, r)
      if (r + 1 < R) RMX_ROW(T1, T2, T0, r + 1)
      if (r + 2 < R) RMX_ROW(T2, T0, T1, r + 2)
    }
#undef RMX_ROW
  }

  // ---- number the flagged pixels: block scan of the per-lane counts --------------------------------
  uint32_t cnt = 0;
#pragma unroll
  for (int q = 0; q < R / 4; ++q) cnt += (uint32_t)__builtin_popcount(Cw[q]);
  uint32_t incl = cnt;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = (uint32_t)__shfl_up((int)incl, o);
    if (lane >= (uint32_t)o) incl += t;
  }
  if (lane == 63) ctl[wv] = incl;
  __syncthreads();
  uint32_t base = incl - cnt, total = 0;
  for (uint32_t k = 0; k < nwv; ++k) {
    const uint32_t sc = ctl[k];
    if (k < wv) base += sc;
    total += sc;
  }

#ifdef RMX_KF_NOSTAGE1  // experiment (tools/): no 5x5 pass
  total = 0;
#endif
  // ---- 5x5 pass over the queue, KF_QCAP candidates at a time ------------------------------------
  for (uint32_t pass0 = 0; pass0 < total; pass0 += KF_QCAP) {
    if (pass0) __syncthreads();  // the previous pass has read its entries
    // entries of this pass: x | tile row << 12 | frame of the block << 17
    if (cnt != 0 && base < pass0 + KF_QCAP && base + cnt > pass0) {
      uint32_t idx = base;
#pragma unroll
      for (int q = 0; q < R / 4; ++q) {
        uint32_t m = Cw[q];
        while (m) {
          const uint32_t b = (uint32_t)__builtin_ctz(m);
          m &= m - 1;
          if (idx >= pass0 && idx < pass0 + KF_QCAP) {
            const uint32_t x = c * 4 + (b >> 3), tr = 2 + 4 * q + (7 - (b & 7u));
            q1[idx - pass0] = x | tr << 12 | j << 17;
          }
          ++idx;
        }
      }
    }
    __syncthreads();
    const uint32_t m1 = min(KF_QCAP, total - pass0);
    for (uint32_t e0 = 0; e0 < m1; e0 += NT) {
      const uint32_t e = e0 + tid;
      const bool act = e < m1;
      const uint32_t ent = q1[act ? e : 0];
      const uint32_t x = ent & 0xfffu, tr = (ent >> 12) & 31u, jj = ent >> 17;
      // window rows tr-2 .. tr+2, columns x-2 .. x+2 out of the key tile
      const uint32_t a0 = jj * p.pitch + (tr - 2) * (uint32_t)W + (x - 2);
      const uint32_t sft = a0 & 3u;
      const uint32_t* wp = reinterpret_cast<const uint32_t*>(tile + (a0 & ~3u));
      uint32_t w[5], ecol[5];
#pragma unroll
      for (int r = 0; r < 5; ++r) {
        const uint32_t lo = wp[0], hi = wp[1];
        wp += p.Wd;
        w[r] = __builtin_amdgcn_alignbyte(hi, lo, sft);  // keys of x-2 .. x+1
        ecol[r] = (hi >> (8 * sft)) & 0xffu;             // key of x+2
      }
      // ranks with bit 7 set (same order; lets the SWAR compare below borrow from it)
      uint32_t v[7];
#pragma unroll
      for (int r = 0; r < 5; ++r) v[r] = ((w[r] >> 4) & 0x0f0f0f0fu) | 0x80808080u;
      {
        const uint32_t e4 = ecol[0] | ecol[1] << 8 | ecol[2] << 16 | ecol[3] << 24;
        v[5] = ((e4 >> 4) & 0x0f0f0f0fu) | 0x80808080u;
        v[6] = (ecol[4] >> 4) | 0x80808080u;  // unused bytes hold rank 0: never >= a threshold >= 1
      }
      const uint32_t p1 = (v[2] >> 16) & 0xfu;
      // p3 = 4th largest of the inner 3x3 (kpe.hpp:313-314), by the network of the 3x3 pass
      uint32_t p3;
      {
        uint32_t lo[3], md[3], hi[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const uint32_t qq = v[r + 1];
          const uint32_t a = (qq >> 8) & 0xffu, b = (qq >> 16) & 0xffu, cc = qq >> 24;
          lo[r] = umin3(a, b, cc), md[r] = umed3(a, b, cc), hi[r] = umax3(a, b, cc);
        }
        const uint32_t a = umax3(lo[0], lo[1], md[2]);
        const uint32_t b = umed3(md[0], md[1], hi[2]);
        const uint32_t cc = umed3(hi[0], hi[1], lo[2]);
        p3 = umed3(a, b, cc) & 0xfu;
      }
      // largest t with #(rank >= t) >= 12 == first bin from the top whose running total reaches
      // 12 (median_pixel(H5, 12), kpe.hpp:326-340)
      uint32_t t = 0;
#pragma unroll
      for (uint32_t bit = 8; bit; bit >>= 1) {
        const uint32_t cth = t | bit;
        const uint32_t T = cth * 0x01010101u;
        uint32_t acc = 0;
#pragma unroll
        for (int i = 0; i < 7; ++i) acc = __builtin_amdgcn_sad_u8((v[i] - T) & 0x80808080u, 0u, acc);
        if ((acc >> 7) >= 12u) t = cth;
      }
      const uint32_t p5 = t;
#ifdef RMX_KF_NOAPPEND  // experiment (tools/): the 5x5 pass without its keypoints
      const bool kp = act && (p3 != p5) && p.kmax == 0xdeadbeefu;
#else
      const bool kp = act && (p3 != p5);
#endif
      if (__builtin_amdgcn_ballot_w64(kp) != 0) {
        const uint32_t weight = (p1 != p5) ? 2u : 1u;
        // code bytes (kpe.hpp:342-379; oracle/remap_oracle.cpp encode_ref): native nibbles
        uint32_t nw[5], ne[5];
#pragma unroll
        for (int r = 0; r < 5; ++r) nw[r] = w[r] & 0x0f0f0f0fu, ne[r] = ecol[r] & 0xfu;
        auto nib16 = [](uint32_t d) {  // [b0|b1<<4, b2|b3<<4] of four nibble-valued bytes
          const uint32_t tt = d | (d >> 4);
          return perm(0u, tt, 0x0c0c0200u);
        };
        const uint32_t c0 = nib16(nw[0]);
        const uint32_t b2 = (nw[1] & 0xfu) | (ne[0] << 4);
        const uint32_t c1 = nib16((nw[1] >> 8) | (ne[1] << 24));
        const uint32_t c2 = nib16(nw[2]);
        const uint32_t b7 = (nw[3] & 0xfu) | (ne[2] << 4);
        const uint32_t c3 = nib16((nw[3] >> 8) | (ne[3] << 24));
        const uint32_t c4 = nib16(nw[4]);
        const uint32_t b12 = weight | (ne[4] << 4);
        if (kp) {
          const uint32_t ff = f0 + jj;
          const uint32_t y = (uint32_t)(y0 - 2) + tr;
          const uint32_t slot = atomicAdd(&p.kcount[ff], 1u);
          if (slot < p.kmax) {
            RawDesc* dd = p.raw + (size_t)ff * p.kmax + slot;
            const uint4 cd = make_uint4(c0 | b2 << 16 | (c1 & 0xffu) << 24, (c1 >> 8) | c2 << 8 | b7 << 24, c3 | c4 << 16, b12);
            const uint4 m4 = make_uint4(x | y << 16, 0u, 0u, 0u);
            reinterpret_cast<uint4*>(dd)[0] = cd;
            reinterpret_cast<uint4*>(dd)[1] = m4;
          } else {
            atomicOr(&p.status[0], 1u);
          }
        }
      }
    }
  }
}

}  // namespace rmx
