// fgs_big.hip — cellular kpm::match (reference kpm.hpp:371-393) for fragments of ANY size.
//
// kpm_cell.hip keeps a pair's tables in LDS: at most 4 096 keypoints per fragment, 1 024
// distinct offsets, 8 192 (offset, cell) entries.  A multi-screen map fragment outgrows that
// (the reference has no limit: its maps live on the heap), so pairs that do not fit are matched
// here with the same algorithm on GLOBAL memory: one workgroup of 1 024 threads per fragment
// (fgb_build) or pair (fgb_count, fgb_vote), every table a slice of an arena the host sizes from
// the keypoint counts and — for the vote tables — from the pair counts fgb_count reports.
//
//   fgb_build   the device analogue of kpr::region (kpr.hpp:93-156) for one fragment: rows
//               grouped by code through an open-addressing table (groups numbered by first
//               occurrence = the order in which region::add creates map nodes; a group's points
//               in scan order, kpr.hpp:121-124), the table doubling as the index for
//               `previous.find(key)` (kpm.hpp:259), and the ITERATION ORDER of the region's
//               std::unordered_map (umap_order_big);
//   fgb_count   per current-map code, in iteration order: the matching previous group and the
//               number of point pairs (kpm.hpp:257-261), prefix-summed: pair index t = first
//               touch order of the totalizator;
//   fgb_vote    votes and first touch per offset, distinct cells per offset (kpm.hpp:231-264),
//               find_best with libstdc++'s tie order (kpm.hpp:281-298), active cells
//               (kpm.hpp:302-367) and the 0.66f test (kpm.hpp:388).
// The same tables serve the grid matcher (kpm.hpp:395-415) where a frame's region outgrows the
// LDS instances of kpm.hip: fgb_rows cuts a region's rows out of a frame's table, fgb_ticket
// writes the region's ticket (rmx_kpm_resolve).
//
// Bit-exactness on ties needs the iteration order of unordered_maps with up to millions of
// nodes: umap_order_big is umap_order_par (kpm.hpp) on global scratch, with libstdc++'s bucket
// growth sequence continued (measured with g++ 11.4: 13, 29, ... 5 967 347).
#include "common.h"

namespace rmx {

constexpr uint32_t FGB_NT = 1024;

__device__ const uint32_t kUmapBucketsBig[19] = {13,     29,     59,     127,    257,     541,     1109,    2357,    5087,   10273,
                                                 20753,  42043,  85229,  172933, 351061,  712697,  1447153, 2938679, 5967347};
constexpr uint32_t FGB_MAX_NODES = 5967347;  // keypoints per fragment / distinct offsets per pair

// smallest bucket count of the growth sequence that holds n nodes (host and device)
__host__ __device__ inline uint32_t fgb_bucket_cap(uint32_t n) {
  const uint32_t b[19] = {13,    29,    59,    127,    257,    541,    1109,    2357,    5087,   10273,
                          20753, 42043, 85229, 172933, 351061, 712697, 1447153, 2938679, 5967347};
  for (int i = 0; i < 19; ++i)
    if (b[i] >= n) return b[i];
  return b[18];
}
__host__ __device__ inline uint32_t fgb_pow2(uint32_t n) {
  uint32_t p = 1;
  while (p < n) p <<= 1;
  return p;
}

// ---- block-wide helpers on global memory (one workgroup, FGB_NT threads) ----------------------------

// ascending bitonic sort of n2 (power of two) 64-bit keys
__device__ inline void fgb_sort(uint64_t* a, uint32_t n2) {
  for (uint32_t k = 2; k <= n2; k <<= 1)
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      __syncthreads();
      for (uint32_t i = threadIdx.x; i < n2; i += FGB_NT) {
        const uint32_t ixj = i ^ j;
        if (ixj > i) {
          const uint64_t x = a[i], y = a[ixj];
          if ((x > y) == ((i & k) == 0)) {
            a[i] = y;
            a[ixj] = x;
          }
        }
      }
    }
  __syncthreads();
}

// exclusive prefix sum of v[0..n) into out[0..n] (out[n] = total), 32-bit; returns the total.
// Every thread sums a contiguous chunk, the chunk sums are scanned through LDS.
__device__ inline uint32_t fgb_exscan(const uint32_t* v, uint32_t* out, uint32_t n, uint32_t* overflow) {
  __shared__ uint32_t s_wave[FGB_NT / 64];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
  const uint32_t per = (n + FGB_NT - 1) / FGB_NT;
  const uint32_t lo = min(n, tid * per), hi = min(n, lo + per);
  uint64_t sum = 0;
  for (uint32_t i = lo; i < hi; ++i) sum += v[i];
  uint64_t incl = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint64_t t = (uint64_t)__shfl_up((long long)incl, o);
    if (lane >= (uint32_t)o) incl += t;
  }
  if (incl > 0x7fffffffull) *overflow = 1;
  if (lane == 63) s_wave[wv] = (uint32_t)incl;
  __syncthreads();
  uint64_t base = 0, total = 0;
  for (uint32_t k = 0; k < FGB_NT / 64; ++k) {
    if (k < wv) base += s_wave[k];
    total += s_wave[k];
  }
  if (total > 0x7fffffffull) *overflow = 1;
  uint32_t run = (uint32_t)(base + incl - sum);
  for (uint32_t i = lo; i < hi; ++i) {
    const uint32_t x = v[i];  // (out may alias v)
    out[i] = run;
    run += x;
  }
  if (tid == 0) out[n] = (uint32_t)total;
  __syncthreads();
  return (uint32_t)total;
}

// Iteration order of a libstdc++ unordered_map after inserting n keys with 64-bit hashes
// h[0..n) in index order: order[k] = index of the k-th iterated key (oracle: orc_umap_order).
// Level by level (kpm.hip: umap_order_par): the list after (re)building with B buckets from a
// sequence S is S sorted by (first position of the element's bucket in S, descending; own
// position in S, descending); S = previous list order followed by the new inserts.
// scratch: cur, nxt [n] u32; keys [pow2 >= n] u64; bmin [fgb_bucket_cap(n)] u32.
__device__ inline void umap_order_big(const uint64_t* h, uint32_t n, uint32_t* cur, uint32_t* nxt, uint64_t* keys, uint32_t* bmin,
                                      uint32_t* order) {
  const uint32_t tid = threadIdx.x;
  uint32_t nk = 0;
  for (int level = 0; level < 19; ++level) {
    const uint32_t B = kUmapBucketsBig[level], mk = min(B, n);
    const uint64_t magic = ~0ull / B;
    auto bucket = [&](uint64_t hv) {
      uint64_t r = hv - __umul64hi(hv, magic) * B;
      while (r >= B) r -= B;
      return (uint32_t)r;
    };
    for (uint32_t b = tid; b < B; b += FGB_NT) bmin[b] = 0xffffffffu;
    __syncthreads();
    for (uint32_t t = tid; t < mk; t += FGB_NT) {
      const uint32_t e = t < nk ? cur[t] : t;
      atomicMin(&bmin[bucket(h[e])], t);
    }
    __syncthreads();
    const uint32_t n2 = fgb_pow2(mk);
    for (uint32_t t = tid; t < n2; t += FGB_NT) {
      uint64_t key = ~0ull;
      if (t < mk) {
        const uint32_t e = t < nk ? cur[t] : t;
        key = (uint64_t)(0xffffffffu - bmin[bucket(h[e])]) << 32 | (uint64_t)(0xffffffffu - t);
      }
      keys[t] = key;
    }
    fgb_sort(keys, n2);
    for (uint32_t j = tid; j < mk; j += FGB_NT) {
      const uint32_t t = 0xffffffffu - (uint32_t)keys[j];
      nxt[j] = t < nk ? cur[t] : t;
    }
    __syncthreads();
    uint32_t* sw = cur;
    cur = nxt;
    nxt = sw;
    nk = mk;
    if (mk == n) break;
  }
  for (uint32_t j = tid; j < n; j += FGB_NT) order[j] = cur[j];
  __syncthreads();
}

// test hook (rmx_debug_umap_order): the order replay on its own
__global__ __launch_bounds__(FGB_NT) void fgb_umap_order(const uint64_t* h, uint32_t n, uint32_t* cur, uint32_t* nxt, uint64_t* keys,
                                                         uint32_t* bmin, uint32_t* order) {
  umap_order_big(h, n, cur, nxt, keys, bmin, order);
}

// ---- per fragment -------------------------------------------------------------------------------------

struct FgbFrag {
  const rmx_desc* desc;  // table; the fragment's K rows in scan order are desc[rows[i]] (rows == null: desc[i])
  const uint32_t* rows;
  const uint8_t* mask;   // blend mask, w x h
  uint32_t K, w, h;
  uint32_t HTS;          // slots of the code table (power of two >= 2 K)
  uint32_t* hdr;         // [0] G, [1] scratch counter
  uint32_t* ht;          // HTS: 0 = empty, else group + 1 (row + 1 while building)
  uint32_t* cnt;         // HTS: rows per slot
  uint32_t* first;       // HTS: smallest row of the slot
  uint32_t* row_slot;    // K
  uint32_t* gfirst;      // K: first row of group g (ascending in g)
  uint32_t* gstart;      // K + 1: start of group g in pts
  uint32_t* pts;         // K: x | y << 16, grouped by code, scan order inside a group
  uint32_t* giter;       // K: group ids in the map's iteration order
  uint32_t* ucur;        // K
  uint32_t* unxt;        // K
  uint32_t* bmin;        // fgb_bucket_cap(K)
  uint64_t* skeys;       // pow2 >= K
  uint64_t* gh;          // K: code hashes by group
};

// Carves arrays out of an arena (16-byte aligned); with base == nullptr it only sizes.
struct FgbCarver {
  uint8_t* base;
  size_t off = 0;
  template <typename T>
  T* take(size_t count) {
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += (count * sizeof(T) + 15) & ~(size_t)15;
    return p;
  }
};
inline void fgb_carve_frag(FgbCarver& c, FgbFrag& f) {
  const size_t k = std::max<uint32_t>(f.K, 1);
  f.HTS = fgb_pow2(std::max<uint32_t>(2 * f.K, 64));
  f.hdr = c.take<uint32_t>(16);
  f.ht = c.take<uint32_t>(f.HTS);
  f.cnt = c.take<uint32_t>(f.HTS);
  f.first = c.take<uint32_t>(f.HTS);
  f.row_slot = c.take<uint32_t>(k);
  f.gfirst = c.take<uint32_t>(k);
  f.gstart = c.take<uint32_t>(k + 1);
  f.pts = c.take<uint32_t>(k);
  f.giter = c.take<uint32_t>(k);
  f.ucur = c.take<uint32_t>(k);
  f.unxt = c.take<uint32_t>(k);
  f.bmin = c.take<uint32_t>(fgb_bucket_cap((uint32_t)k));
  f.skeys = c.take<uint64_t>(fgb_pow2((uint32_t)k));
  f.gh = c.take<uint64_t>(k);
}

__global__ __launch_bounds__(FGB_NT) void fgb_build(const FgbFrag* frags) {
  const FgbFrag F = frags[blockIdx.x];
  const uint32_t tid = threadIdx.x, K = F.K, hmask = F.HTS - 1;
  auto row = [&](uint32_t i) { return F.rows ? F.rows[i] : i; };
  for (uint32_t s = tid; s < F.HTS; s += FGB_NT) {
    F.ht[s] = 0;
    F.cnt[s] = 0;
    F.first[s] = 0xffffffffu;
  }
  if (tid < 4) F.hdr[tid] = 0;
  __syncthreads();
  if (K == 0) return;
  // rows -> slots (one slot per distinct code)
  for (uint32_t i = tid; i < K; i += FGB_NT) {
    uint32_t c[4];
    load_code(F.desc + row(i), c);
    uint32_t s = fnv_code_lo(c) & hmask;
    for (;;) {
      const uint32_t v = atomicCAS(&F.ht[s], 0u, i + 1);
      if (v == 0u) break;
      uint32_t c2[4];
      load_code(F.desc + row(v - 1), c2);
      if (same_code(c, c2)) break;
      s = (s + 1) & hmask;
    }
    F.row_slot[i] = s;
    atomicAdd(&F.cnt[s], 1u);
    atomicMin(&F.first[s], i);
  }
  __syncthreads();
  // groups in first-occurrence order
  for (uint32_t s = tid; s < F.HTS; s += FGB_NT)
    if (F.ht[s] != 0) F.skeys[atomicAdd(&F.hdr[1], 1u)] = (uint64_t)F.first[s] << 32 | s;
  __syncthreads();
  const uint32_t G = F.hdr[1];
  const uint32_t g2 = fgb_pow2(G);
  for (uint32_t i = G + tid; i < g2; i += FGB_NT) F.skeys[i] = ~0ull;
  fgb_sort(F.skeys, g2);
  for (uint32_t g = tid; g < G; g += FGB_NT) {
    const uint32_t s = (uint32_t)F.skeys[g];
    F.gfirst[g] = row((uint32_t)(F.skeys[g] >> 32));  // the table row that holds the group's code
    F.ht[s] = g + 1;
    F.unxt[g] = F.cnt[s];  // group sizes, scanned below
  }
  __syncthreads();
  fgb_exscan(F.unxt, F.gstart, G, &F.hdr[2]);
  // points: rows sorted by (group, row)
  const uint32_t k2 = fgb_pow2(K);
  for (uint32_t i = tid; i < k2; i += FGB_NT) F.skeys[i] = i < K ? (uint64_t)(F.ht[F.row_slot[i]] - 1) << 32 | i : ~0ull;
  fgb_sort(F.skeys, k2);
  for (uint32_t q = tid; q < K; q += FGB_NT) F.pts[q] = load_xy(F.desc + row((uint32_t)F.skeys[q]));
  // iteration order of the code map
  for (uint32_t g = tid; g < G; g += FGB_NT) {
    uint32_t c[4];
    load_code(F.desc + F.gfirst[g], c);
    F.gh[g] = fnv_code(c);
  }
  __syncthreads();
  umap_order_big(F.gh, G, F.ucur, F.unxt, F.skeys, F.bmin, F.giter);
  if (tid == 0) F.hdr[0] = G;
}

// The rows of ONE grid region of a frame's table (kpr::grid region r, kpe.hpp:157-192), in
// scan order, for the grid matcher's global-memory instance: rows[0..K) and {K, weight-2 rows,
// weight-1 rows} (kpr.hpp:121-124 weight counters).  One workgroup per (table, region).
struct FgbRegion {
  const rmx_desc* desc;
  uint32_t k, region;
  uint32_t* rows;   // k entries
  uint32_t* info;   // [0] K, [1] weight-2 rows, [2] weight-1 rows
};
__global__ __launch_bounds__(FGB_NT) void fgb_rows(const FgbRegion* regs) {
  __shared__ uint32_t s_wave[FGB_NT / 64];
  __shared__ uint32_t s_w[2];
  const FgbRegion R = regs[blockIdx.x];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
  if (tid < 2) s_w[tid] = 0;
  const uint32_t per = (R.k + FGB_NT - 1) / FGB_NT;
  const uint32_t lo = min(R.k, tid * per), hi = min(R.k, lo + per);
  uint32_t cnt = 0, w2 = 0, w1 = 0;
  for (uint32_t i = lo; i < hi; ++i)
    if ((R.desc[i].region_mask >> R.region) & 1u) {
      ++cnt;
      const uint32_t wgt = R.desc[i].code[12] & 0xfu;
      w2 += wgt == 2;
      w1 += wgt == 1;
    }
  uint32_t incl = cnt;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const uint32_t t = (uint32_t)__shfl_up((int)incl, o);
    if (lane >= (uint32_t)o) incl += t;
  }
  if (lane == 63) s_wave[wv] = incl;
  __syncthreads();
  uint32_t base = 0, total = 0;
  for (uint32_t k = 0; k < FGB_NT / 64; ++k) {
    if (k < wv) base += s_wave[k];
    total += s_wave[k];
  }
  uint32_t at = base + incl - cnt;
  for (uint32_t i = lo; i < hi; ++i)
    if ((R.desc[i].region_mask >> R.region) & 1u) R.rows[at++] = i;
  if (w2) atomicAdd(&s_w[0], w2);
  if (w1) atomicAdd(&s_w[1], w1);
  __syncthreads();
  if (tid == 0) {
    R.info[0] = total;
    R.info[1] = s_w[0];
    R.info[2] = s_w[1];
  }
}

// ---- per pair -----------------------------------------------------------------------------------------

struct FgbPair {
  uint32_t fp, fc;   // previous, current fragment (indices into the FgbFrag array)
  uint32_t out;      // index of the pair in the caller's vote array (grid matcher: ticket index)
  uint32_t w2_only;  // grid matcher, kpm.hpp:112-117: only codes of weight 2 vote (both regions rich in them)
  uint32_t HS, PS;   // slots of the offset table / of the (offset, cell) set (powers of two)
  uint32_t Dcap;     // distinct offsets the tie scratch holds
  // per current code, in iteration order (Kc + 1 entries): previous group and pairs before it
  uint32_t* gpk;
  uint32_t* base;
  uint32_t* total;   // [0] point pairs of the pair, [1] overflow
  // vote tables (sized after fgb_count)
  uint32_t* hk;      // HS: offset key, 0 = empty
  uint32_t* hc;      // HS: votes
  uint32_t* hs;      // HS: first pair index
  uint32_t* hcell;   // HS: distinct cells
  uint64_t* pc;      // PS: (offset slot + 1) << 32 | cell key
  // tie scratch (Dcap offsets)
  uint64_t* skeys;   // pow2 >= Dcap
  uint64_t* hh;      // Dcap
  uint32_t* ucur;    // Dcap
  uint32_t* unxt;    // Dcap
  uint32_t* pos_of;  // Dcap
  uint32_t* bmin;    // fgb_bucket_cap(Dcap)
  uint32_t* ctl;     // 8 words
};

// counting arrays (known from the current fragment's K) and, once the pair count is known, the
// vote tables
inline void fgb_carve_count(FgbCarver& c, FgbPair& q, uint32_t Kc) {
  q.gpk = c.take<uint32_t>((size_t)Kc + 1);
  q.base = c.take<uint32_t>((size_t)Kc + 1);
  q.total = c.take<uint32_t>(4);
}
// (kc_cells: the current fragment's keypoints when the (offset, cell) set also has to hold the
// active cells of fgb_vote — at most one per keypoint; 0 for fgb_ticket)
inline void fgb_carve_vote(FgbCarver& c, FgbPair& q, uint32_t total, uint32_t kc_cells) {
  const uint32_t d = std::min<uint32_t>(std::max<uint32_t>(total, 1), FGB_MAX_NODES);
  q.Dcap = d;
  q.HS = fgb_pow2(std::max<uint32_t>(2 * d, 64));
  q.PS = fgb_pow2(std::max<uint32_t>(2 * std::max<uint32_t>(std::max<uint32_t>(total, kc_cells), 32), 64));
  q.hk = c.take<uint32_t>(q.HS);
  q.hc = c.take<uint32_t>(q.HS);
  q.hs = c.take<uint32_t>(q.HS);
  q.hcell = c.take<uint32_t>(q.HS);
  q.pc = c.take<uint64_t>(q.PS);
  q.skeys = c.take<uint64_t>(fgb_pow2(d));
  q.hh = c.take<uint64_t>(d);
  q.ucur = c.take<uint32_t>(d);
  q.unxt = c.take<uint32_t>(d);
  q.pos_of = c.take<uint32_t>(d);
  q.bmin = c.take<uint32_t>(fgb_bucket_cap(d));
  q.ctl = c.take<uint32_t>(8);
}

__global__ __launch_bounds__(FGB_NT) void fgb_count(const FgbFrag* frags, const FgbPair* pairs) {
  const FgbPair Q = pairs[blockIdx.x];
  const FgbFrag P = frags[Q.fp], C = frags[Q.fc];
  const uint32_t tid = threadIdx.x;
  const uint32_t Gc = C.K ? C.hdr[0] : 0, Gp = P.K ? P.hdr[0] : 0;
  if (tid < 2) Q.total[tid] = 0;
  __syncthreads();
  if (Gc == 0 || Gp == 0) return;
  const uint32_t pmask = P.HTS - 1;
  for (uint32_t k = tid; k < Gc; k += FGB_NT) {
    const uint32_t gc = C.giter[k];
    uint32_t c[4];
    load_code(C.desc + C.gfirst[gc], c);
    uint32_t s = fnv_code_lo(c) & pmask, np = 0, gp = 0;
    for (;;) {
      const uint32_t v = P.ht[s];
      if (v == 0) break;
      uint32_t c2[4];
      load_code(P.desc + P.gfirst[v - 1], c2);
      if (same_code(c, c2)) {
        gp = v - 1;
        np = P.gstart[gp + 1] - P.gstart[gp];
        break;
      }
      s = (s + 1) & pmask;
    }
    if (Q.w2_only && (c[3] & 0xfu) != 2u) np = 0;
    const uint64_t pr = (uint64_t)np * (C.gstart[gc + 1] - C.gstart[gc]);
    if (pr > 0x7fffffffull) Q.total[1] = 1;
    Q.gpk[k] = gp;
    Q.base[k] = (uint32_t)pr;
  }
  __syncthreads();
  const uint32_t total = fgb_exscan(Q.base, Q.base, Gc, &Q.total[1]);
  if (tid == 0) Q.total[0] = total;
}

__device__ __forceinline__ uint64_t fgb_max_u64(uint64_t v, uint64_t* s_red) {
  const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const uint64_t t = (uint64_t)__shfl_xor((long long)v, o);
    v = t > v ? t : v;
  }
  __syncthreads();
  if (lane == 0) s_red[wv] = v;
  __syncthreads();
  uint64_t m = 0;
  for (uint32_t k = 0; k < FGB_NT / 64; ++k) m = s_red[k] > m ? s_red[k] : m;
  return m;
}
__device__ __forceinline__ uint32_t fgb_sum_u32(uint32_t v, uint32_t* s_red) {
  const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += (uint32_t)__shfl_xor((int)v, o);
  __syncthreads();
  if (lane == 0) s_red[wv] = v;
  __syncthreads();
  uint32_t m = 0;
  for (uint32_t k = 0; k < FGB_NT / 64; ++k) m += s_red[k];
  return m;
}

__global__ __launch_bounds__(FGB_NT) void fgb_vote(const FgbFrag* frags, const FgbPair* pairs, rmx_vote* votes, uint32_t cell_w, uint32_t cell_h,
                                                   uint32_t* status) {
  __shared__ uint64_t s_red64[FGB_NT / 64];
  __shared__ uint32_t s_red32[FGB_NT / 64];
  const FgbPair Q = pairs[blockIdx.x];
  const FgbFrag P = frags[Q.fp], C = frags[Q.fc];
  const uint32_t tid = threadIdx.x;
  rmx_vote* out = votes + Q.out;
  if (tid == 0) *out = rmx_vote{0, 0, 0, 0};
  const uint32_t total = Q.total[0];
  if (total == 0) return;  // offsets.empty() -> nullopt (kpm.hpp:380-382)
  const uint32_t Gc = C.hdr[0];
  for (uint32_t s = tid; s < Q.HS; s += FGB_NT) {
    Q.hk[s] = 0;
    Q.hc[s] = 0;
    Q.hs[s] = 0xffffffffu;
    Q.hcell[s] = 0;
  }
  for (uint32_t s = tid; s < Q.PS; s += FGB_NT) Q.pc[s] = 0ull;
  if (tid < 8) Q.ctl[tid] = 0;
  __syncthreads();

  // ---- votes per offset and distinct cells per offset (kpm.hpp:231-264) ----
  uint32_t my_new = 0, my_fail = 0;
  const uint32_t hmask = Q.HS - 1, pmask2 = Q.PS - 1;
  for (uint32_t t = tid; t < total; t += FGB_NT) {
    uint32_t lo = 0, hi = Gc;  // last k with base[k] <= t (its code has pairs: base[k + 1] > t)
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (Q.base[mid] <= t)
        lo = mid;
      else
        hi = mid;
    }
    const uint32_t gc = C.giter[lo], gp = Q.gpk[lo], rel = t - Q.base[lo];
    const uint32_t cs = C.gstart[gc], nc = C.gstart[gc + 1] - cs;
    const uint32_t ip = rel / nc, ic = rel - ip * nc;
    const uint32_t pp = P.pts[P.gstart[gp] + ip], cp = C.pts[cs + ic];
    const int32_t px = (int32_t)(pp & 0xffffu), py = (int32_t)(pp >> 16), cx = (int32_t)(cp & 0xffffu), cy = (int32_t)(cp >> 16);
    const uint32_t key = off_key(px - cx, py - cy);
    const uint32_t cellk = (uint32_t)(min(px, cx) / (int32_t)cell_w) << 12 | (uint32_t)(min(py, cy) / (int32_t)cell_h);
    uint32_t s = (key * 2654435761u) & hmask;
    bool placed = false;
    for (uint32_t probes = 0; probes <= hmask; ++probes) {
      const uint32_t v = atomicCAS(&Q.hk[s], 0u, key);
      if (v == 0u) ++my_new;
      if (v == 0u || v == key) {
        atomicAdd(&Q.hc[s], 1u);
        atomicMin(&Q.hs[s], t);
        placed = true;
        break;
      }
      s = (s + 1) & hmask;
    }
    if (!placed) {
      ++my_fail;
      continue;
    }
    const unsigned long long k64 = (unsigned long long)(s + 1) << 32 | cellk;
    uint32_t h = (uint32_t)((k64 * 0x9E3779B97F4A7C15ull) >> 32) & pmask2;
    bool done = false;
    for (uint32_t probes = 0; probes <= pmask2; ++probes) {
      const unsigned long long v = atomicCAS(&Q.pc[h], 0ull, k64);
      if (v == 0ull) atomicAdd(&Q.hcell[s], 1u);
      if (v == 0ull || v == k64) {
        done = true;
        break;
      }
      h = (h + 1) & pmask2;
    }
    if (!done) ++my_fail;
  }
  const uint32_t D = fgb_sum_u32(my_new, s_red32);
  const uint32_t Fl = fgb_sum_u32(my_fail, s_red32);
  if (Fl != 0 || D > Q.Dcap) {
    if (tid == 0) atomicOr(&status[1], 16u);
    return;
  }

  // ---- find_best: most votes, first in the totalizator's iteration order on ties (kpm.hpp:281-298) ----
  uint64_t m = 0;
  for (uint32_t s = tid; s < Q.HS; s += FGB_NT)
    if (Q.hk[s] != 0) {
      const uint64_t cand = (uint64_t)Q.hc[s] << 32 | (uint64_t)(0xffffffffu - s);
      m = cand > m ? cand : m;
    }
  const uint64_t best = fgb_max_u64(m, s_red64);
  const uint32_t cmax = (uint32_t)(best >> 32);
  uint32_t best_slot = 0xffffffffu - (uint32_t)best;
  uint32_t ties = 0;
  for (uint32_t s = tid; s < Q.HS; s += FGB_NT) ties += (Q.hk[s] != 0 && Q.hc[s] == cmax) ? 1u : 0u;
  ties = fgb_sum_u32(ties, s_red32);
  if (ties > 1) {
    // replay the totalizator's node order: insertion = first touch
    for (uint32_t s = tid; s < Q.HS; s += FGB_NT)
      if (Q.hk[s] != 0) Q.skeys[atomicAdd(&Q.ctl[0], 1u)] = (uint64_t)Q.hs[s] << 32 | s;
    __syncthreads();
    const uint32_t d2 = fgb_pow2(D);
    for (uint32_t i = D + tid; i < d2; i += FGB_NT) Q.skeys[i] = ~0ull;
    fgb_sort(Q.skeys, d2);
    uint32_t* slot_of = Q.pos_of;  // slot of the i-th inserted offset
    for (uint32_t i = tid; i < D; i += FGB_NT) {
      const uint32_t sl = (uint32_t)Q.skeys[i], key = Q.hk[sl];
      Q.hh[i] = fnv_offset((int32_t)(key >> 16) - 32768, (int32_t)(key & 0xffffu) - 32768);
      slot_of[i] = sl;
    }
    __syncthreads();
    // order[k] = insertion index of the k-th iterated offset (written over the order scratch)
    uint32_t* order = Q.ucur;
    umap_order_big(Q.hh, D, Q.ucur, Q.unxt, Q.skeys, Q.bmin, order);
    uint32_t firstpos = 0xffffffffu;  // smallest iteration position holding cmax votes
    for (uint32_t k = tid; k < D; k += FGB_NT)
      if (Q.hc[slot_of[order[k]]] == cmax) firstpos = min(firstpos, k);
    const uint64_t mm = fgb_max_u64((uint64_t)(0xffffffffu - firstpos), s_red64);
    best_slot = slot_of[order[0xffffffffu - (uint32_t)mm]];
  }
  const uint32_t bkey = Q.hk[best_slot], matched_cells = Q.hcell[best_slot];
  const int32_t bdx = (int32_t)(bkey >> 16) - 32768, bdy = (int32_t)(bkey & 0xffffu) - 32768;

  // ---- active cells (kpm.hpp:302-367; only the current-side limits are used) ----
  const uint32_t pw = P.w, ph = P.h, cw = C.w, chh = C.h;
  auto limits = [](int32_t delta, uint64_t previous, uint64_t current, uint64_t& lo, uint64_t& hi) {
    if (delta < 0) {
      const uint64_t d = (uint64_t)(-(int64_t)delta);
      lo = d;
      hi = min(current, previous + d);
    } else {
      lo = 0;
      hi = min(current, previous - (uint64_t)delta);  // wraps like the reference's size_t arithmetic
    }
  };
  uint64_t left, right, top, bottom;
  limits(bdx, pw, cw, left, right);
  limits(bdy, ph, chh, top, bottom);
  __syncthreads();
  for (uint32_t s = tid; s < Q.PS; s += FGB_NT) Q.pc[s] = 0ull;
  __syncthreads();
  uint32_t act = 0, fail2 = 0;
  for (uint32_t i = tid; i < C.K; i += FGB_NT) {
    const uint32_t cp = C.pts[i];
    const uint64_t x = cp & 0xffffu, y = cp >> 16;
    if (x >= left && x < right && y >= top && y < bottom) {
      const int64_t idx = (int64_t)pw * ((int64_t)y + bdy) + ((int64_t)x + bdx);
      if (P.mask[idx] != 0) {
        const uint32_t ox = (uint32_t)((int32_t)(x - left) / (int32_t)cell_w), oy = (uint32_t)((int32_t)(y - top) / (int32_t)cell_h);
        const unsigned long long k64 = 1ull << 32 | (unsigned long long)(ox << 12 | oy);
        uint32_t h = (uint32_t)((k64 * 0x9E3779B97F4A7C15ull) >> 32) & pmask2;
        bool done = false;
        for (uint32_t probes = 0; probes <= pmask2; ++probes) {
          const unsigned long long v = atomicCAS(&Q.pc[h], 0ull, k64);
          if (v == 0ull) ++act;
          if (v == 0ull || v == k64) {
            done = true;
            break;
          }
          h = (h + 1) & pmask2;
        }
        if (!done) ++fail2;
      }
    }
  }
  act = fgb_sum_u32(act, s_red32);
  fail2 = fgb_sum_u32(fail2, s_red32);
  if (fail2) {
    if (tid == 0) atomicOr(&status[1], 32u);
    return;
  }
  // ---- kpm.hpp:388 ----
  if (tid == 0) {
    const float lhs = (float)matched_cells, rhs = (float)act * 0.66f;
    if (!(lhs < rhs)) *out = rmx_vote{bdx, bdy, cmax, 1u};
  }
}

// Grid matcher (kpm.hpp:395-415), one (pair, region): cast_vote = top_offsets(count_offsets, 3)
// (kpm.hpp:105-159) — the three offsets with most votes, equal counts in the totalizator's
// iteration order — written as the ticket kpm_declare reads: {key, count} x 3, key 0 ends it.
__global__ __launch_bounds__(FGB_NT) void fgb_ticket(const FgbFrag* frags, const FgbPair* pairs, uint32_t* tickets, uint32_t* status) {
  __shared__ uint64_t s_red64[FGB_NT / 64];
  __shared__ uint32_t s_red32[FGB_NT / 64];
  const FgbPair Q = pairs[blockIdx.x];
  const FgbFrag P = frags[Q.fp], C = frags[Q.fc];
  const uint32_t tid = threadIdx.x;
  uint32_t* ticket = tickets + (size_t)Q.out * 6;
  if (tid < 6) ticket[tid] = 0;
  const uint32_t total = Q.total[0];
  if (total == 0) return;
  const uint32_t Gc = C.hdr[0];
  for (uint32_t s = tid; s < Q.HS; s += FGB_NT) {
    Q.hk[s] = 0;
    Q.hc[s] = 0;
    Q.hs[s] = 0xffffffffu;
  }
  if (tid < 8) Q.ctl[tid] = 0;
  __syncthreads();
  uint32_t my_new = 0, my_fail = 0;
  const uint32_t hmask = Q.HS - 1;
  for (uint32_t t = tid; t < total; t += FGB_NT) {
    uint32_t lo = 0, hi = Gc;
    while (hi - lo > 1) {
      const uint32_t mid = (lo + hi) >> 1;
      if (Q.base[mid] <= t)
        lo = mid;
      else
        hi = mid;
    }
    const uint32_t gc = C.giter[lo], gp = Q.gpk[lo], rel = t - Q.base[lo];
    const uint32_t cs = C.gstart[gc], nc = C.gstart[gc + 1] - cs;
    const uint32_t ip = rel / nc, ic = rel - ip * nc;
    const uint32_t pp = P.pts[P.gstart[gp] + ip], cp = C.pts[cs + ic];
    const uint32_t key = off_key((int32_t)(pp & 0xffffu) - (int32_t)(cp & 0xffffu), (int32_t)(pp >> 16) - (int32_t)(cp >> 16));
    uint32_t s = (key * 2654435761u) & hmask;
    bool placed = false;
    for (uint32_t probes = 0; probes <= hmask; ++probes) {
      const uint32_t v = atomicCAS(&Q.hk[s], 0u, key);
      if (v == 0u) ++my_new;
      if (v == 0u || v == key) {
        atomicAdd(&Q.hc[s], 1u);
        atomicMin(&Q.hs[s], t);
        placed = true;
        break;
      }
      s = (s + 1) & hmask;
    }
    if (!placed) ++my_fail;
  }
  const uint32_t D = fgb_sum_u32(my_new, s_red32);
  const uint32_t Fl = fgb_sum_u32(my_fail, s_red32);
  if (Fl != 0 || D > Q.Dcap) {
    if (tid == 0) atomicOr(&status[1], 16u);
    return;
  }
  // the totalizator's nodes in insertion (= first touch) order, then in iteration order
  for (uint32_t s = tid; s < Q.HS; s += FGB_NT)
    if (Q.hk[s] != 0) Q.skeys[atomicAdd(&Q.ctl[0], 1u)] = (uint64_t)Q.hs[s] << 32 | s;
  __syncthreads();
  const uint32_t d2 = fgb_pow2(D);
  for (uint32_t i = D + tid; i < d2; i += FGB_NT) Q.skeys[i] = ~0ull;
  fgb_sort(Q.skeys, d2);
  uint32_t* slot_of = Q.pos_of;
  for (uint32_t i = tid; i < D; i += FGB_NT) {
    const uint32_t sl = (uint32_t)Q.skeys[i], key = Q.hk[sl];
    Q.hh[i] = fnv_offset((int32_t)(key >> 16) - 32768, (int32_t)(key & 0xffffu) - 32768);
    slot_of[i] = sl;
  }
  __syncthreads();
  uint32_t* order = Q.ucur;
  umap_order_big(Q.hh, D, Q.ucur, Q.unxt, Q.skeys, Q.bmin, order);
  // top_offsets(total, 3): most votes first, equal counts in iteration order (strict >, kpm.hpp:138)
  uint64_t taken[3] = {0, 0, 0};
  for (int rk = 0; rk < 3 && (uint32_t)rk < D; ++rk) {
    uint64_t m = 0;
    for (uint32_t k = tid; k < D; k += FGB_NT) {
      const uint64_t cand = (uint64_t)Q.hc[slot_of[order[k]]] << 32 | (uint64_t)(0xffffffffu - k);
      bool used = false;
      for (int j = 0; j < rk; ++j) used |= cand == taken[j];
      if (!used && cand > m) m = cand;
    }
    taken[rk] = fgb_max_u64(m, s_red64);
    if (tid == 0) {
      const uint32_t sl = slot_of[order[0xffffffffu - (uint32_t)taken[rk]]];
      ticket[2 * rk] = Q.hk[sl];
      ticket[2 * rk + 1] = Q.hc[sl];
    }
  }
}

}  // namespace rmx
