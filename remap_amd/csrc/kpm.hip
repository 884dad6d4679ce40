// kpm.hip — keypoint matching on gfx950: grid/Borda variant (reference kpm.hpp:395-415).
//
// Three kernels per batch of descriptor tables:
//
//   kpm_build   one workgroup per frame.  Builds the device analogue of kpr::grid
//               (kpr.hpp:93-223): for every region the keypoints grouped by code (groups in
//               first-occurrence order = the order in which region::add creates map nodes,
//               points of a group in scan order = push_back order, kpr.hpp:121-124), a hash
//               index code -> group for the `previous.find(key)` of kpm.hpp:119, the weight
//               counters, and the ITERATION ORDER of the region's std::unordered_map
//               (what `for (auto& [key, curr] : current.points())`, kpm.hpp:112, walks).
//   kpm_match   one wave per (frame pair, region) = cast_vote (kpm.hpp:213-223):
//               count_offsets (all point pairs of every shared code, kpm.hpp:91-125) into
//               an LDS hash of offsets, then top_offsets(…, 3) (kpm.hpp:127-159).
//   kpm_declare one thread per pair: Borda count + declare (kpm.hpp:172-211).
//
// Bit-exactness.  top_offsets keeps the FIRST-ITERATED offset among equal counts
// (kpm.hpp:138, strict >), iterating a std::unordered_map, so the ticket depends on
// libstdc++'s node order.  That order is a deterministic function of the insertion
// sequence and the hashes: umap_order_seq() / umap_order_par() below replay libstdc++'s
// _M_insert_bucket_begin / _M_rehash_aux (GCC 11) on the device.  It needs the offsets in
// the order the reference first touches them, i.e. in pair-enumeration order, which in
// turn follows the iteration order of the current region's code map — hence the two
// levels of emulation.  The replay only runs when the top-3 actually contains a tie; the
// final declare() is order-independent given exact tickets.
#include "common.h"

namespace rmx {

// ---- hashes of the reference ------------------------------------------------------------------
// kpr::code_hash, kpr.hpp:29-38 (64-bit size_t arithmetic)
__device__ __forceinline__ uint64_t fnv_code(const uint32_t c[4]) {
  uint64_t h = 2166136261ull;
#pragma unroll
  for (int i = 0; i < 13; ++i) {
    h ^= (uint64_t)((c[i >> 2] >> (8 * (i & 3))) & 0xffu);
    h *= 16777619ull;
  }
  return h;
}
// its low 32 bits (all a slot index needs): the same recurrence in 32-bit arithmetic
__device__ __forceinline__ uint32_t fnv_code_lo(const uint32_t c[4]) {
  uint32_t h = 2166136261u;
#pragma unroll
  for (int i = 0; i < 13; ++i) {
    h ^= (c[i >> 2] >> (8 * (i & 3))) & 0xffu;
    h *= 16777619u;
  }
  return h;
}
// cdt::offset_hash, cdt.hpp:158-171 (int32 -> size_t sign-extends)
__device__ __forceinline__ uint64_t fnv_offset(int32_t x, int32_t y) {
  uint64_t h = 2166136261ull;
  h ^= (uint64_t)(int64_t)x;
  h *= 16777619ull;
  h ^= (uint64_t)(int64_t)y;
  h *= 16777619ull;
  return h;
}

// libstdc++ _Prime_rehash_policy growth sequence for a default-constructed unordered_map
// filled one element at a time: the table is rehashed to kUmapBuckets[l] when the element
// count reaches the previous bucket count (measured with g++ 11.4, see
// tests/test_umap_order.py; hashtable_c++0x.cc _M_need_rehash/_M_next_bkt).
__device__ const uint32_t kUmapBuckets[12] = {13, 29, 59, 127, 257, 541, 1109, 2357, 5087, 10273, 20753, 42043};
// floor((2^64 - 1) / bucket count): Barrett reduction instead of a software 64-bit division
__device__ const uint64_t kUmapMagic[12] = {0x13b13b13b13b13b1ull, 0x8d3dcb08d3dcb08ull, 0x456c797dd49c341ull, 0x204081020408102ull, 0xff00ff00ff00ffull, 0x79237d65bcce50ull, 0x3b183cf0fed886ull, 0x1bce09c66f6fc3ull, 0xce20e98148847ull, 0x661230fdfb43eull, 0x3286c744cba2bull, 0x18f0c8989d637ull};

// h mod kUmapBuckets[level], exact: q = mulhi(h, M) is floor(h / B) or up to 2 less.
__device__ __forceinline__ uint32_t umap_bucket(uint64_t h, int level) {
  const uint32_t B = kUmapBuckets[level];
  uint64_t r = h - __umul64hi(h, kUmapMagic[level]) * B;
  while (r >= B) r -= B;
  return (uint32_t)r;
}

// Iteration order of a libstdc++ unordered_map after inserting keys with 64-bit hashes
// h[0..n-1] in index order (oracle: orc_umap_order).  Sequential; called by ONE lane.
// next/bucket: 16-bit scratch (next: n entries, bucket: >= final bucket count entries).
// order[k] = index of the k-th iterated key.
template <typename OutT>
__device__ inline void umap_order_seq(const uint64_t* h, uint32_t n, uint16_t* next, uint16_t* bucket, OutT* order) {
  const uint32_t NONE = 0xffffu, BEFORE = 0xfffeu;
  uint32_t head = NONE, B = 1;
  int level = -1;
  for (uint32_t i = 0; i < n; ++i) {
    if (i == 0 || i == B) {
      ++level;
      const uint32_t Bn = kUmapBuckets[level];
      for (uint32_t b = 0; b < Bn; ++b) bucket[b] = (uint16_t)NONE;
      uint32_t p = head, bbegin = 0;
      head = NONE;
      while (p != NONE) {  // _M_rehash_aux
        uint32_t nx = next[p];
        uint32_t b = umap_bucket(h[p], level);
        uint32_t before = bucket[b];
        if (before == NONE) {
          next[p] = (uint16_t)head;
          head = p;
          bucket[b] = (uint16_t)BEFORE;
          if (next[p] != NONE) bucket[bbegin] = (uint16_t)p;
          bbegin = b;
        } else if (before == BEFORE) {
          next[p] = (uint16_t)head;
          head = p;
        } else {
          next[p] = next[before];
          next[before] = (uint16_t)p;
        }
        p = nx;
      }
      B = Bn;
    }
    // _M_insert_bucket_begin
    uint32_t b = umap_bucket(h[i], level);
    uint32_t before = bucket[b];
    if (before == NONE) {
      next[i] = (uint16_t)head;
      head = i;
      if (next[i] != NONE) bucket[umap_bucket(h[next[i]], level)] = (uint16_t)i;
      bucket[b] = (uint16_t)BEFORE;
    } else if (before == BEFORE) {
      next[i] = (uint16_t)head;
      head = i;
    } else {
      next[i] = next[before];
      next[before] = (uint16_t)i;
    }
  }
  uint32_t k = 0;
  for (uint32_t p = head; p != NONE; p = next[p]) order[k++] = (OutT)p;
}

// The same order without pointer chasing.  libstdc++ puts a node at the front of its bucket
// and a new bucket at the front of the list, for inserts and for rehashes alike, so the list
// after (re)building with B buckets from a sequence S is S sorted by (first position of the
// element's bucket in S, descending; own position in S, descending).  Level by level — S =
// previous list order followed by the elements inserted while that bucket count was live —
// this is a handful of parallel passes (bucket index, atomicMin per bucket, bitonic sort)
// instead of thousands of dependent LDS round trips.  Verified against a real
// std::unordered_map through the oracle (tests/test_host_logic.py restates it in numpy).
//   NT threads cooperate (a wave with a wave barrier, or a workgroup with __syncthreads);
//   scratch: keys[pow2 >= n] u32, seq_a/seq_b[n] u16, bmin[final bucket count] u32.
struct WaveBarrier {
  __device__ __forceinline__ void operator()() const {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
  }
};
struct BlockBarrier {
  __device__ __forceinline__ void operator()() const { __syncthreads(); }
};

// one level of umap_order_par held by ONE wave in registers: element t in slot t >> 6 of lane
// t & 63; the bucket index (a Barrett reduction of the 64-bit hash) is computed once
template <uint32_t KPL, typename Bar>
__device__ __forceinline__ void umap_level_regs(const uint64_t* h, int level, uint32_t mk, uint32_t nk, uint32_t n2,
                                                const uint16_t* cur, uint16_t* nxt, uint32_t* bmin, Bar bar, uint32_t lane) {
  uint32_t bkt[KPL];
#pragma unroll
  for (uint32_t sl = 0; sl < KPL; ++sl) {
    const uint32_t t = sl * 64 + lane;
    bkt[sl] = 0;
    if (t < mk) {
      const uint32_t e = t < nk ? cur[t] : t;
      bkt[sl] = umap_bucket(h[e], level);
      atomicMin(&bmin[bkt[sl]], t);
    }
  }
  bar();
  uint32_t key[KPL];
#pragma unroll
  for (uint32_t sl = 0; sl < KPL; ++sl) {
    const uint32_t t = sl * 64 + lane;
    key[sl] = t < mk ? (0xffffu - bmin[bkt[sl]]) << 16 | (0xffffu - t) : 0xffffffffu;
  }
  wave_sort<KPL>(key, n2, lane);
#pragma unroll
  for (uint32_t sl = 0; sl < KPL; ++sl) {
    const uint32_t j = sl * 64 + lane;
    if (j < mk) {
      const uint32_t t = 0xffffu - (key[sl] & 0xffffu);
      nxt[j] = (uint16_t)(t < nk ? cur[t] : t);
    }
  }
  bar();
}

template <uint32_t NT, uint32_t NMAX = 0xffffffffu, typename Bar, typename OutT>
__device__ inline void umap_order_par(const uint64_t* h, uint32_t n, uint32_t* keys, uint16_t* seq_a, uint16_t* seq_b,
                                      uint32_t* bmin, Bar bar, OutT* order, uint32_t tid) {
  uint16_t* cur = seq_a;
  uint16_t* nxt = seq_b;
  uint32_t nk = 0;
  for (int level = 0;; ++level) {
    const uint32_t B = kUmapBuckets[level], mk = min(B, n);
    for (uint32_t b = tid; b < B; b += NT) bmin[b] = 0xffffffffu;
    bar();
    uint32_t n2 = 1;
    while (n2 < mk) n2 <<= 1;
    if (NT == 64 && n2 <= (NMAX > 256 ? 512u : NMAX > 64 ? 256u : 64u)) {
      if (n2 <= 64)
        umap_level_regs<1>(h, level, mk, nk, n2, cur, nxt, bmin, bar, tid);
      else if (NMAX > 64 && n2 <= 256)
        umap_level_regs<4>(h, level, mk, nk, n2, cur, nxt, bmin, bar, tid);
      else if (NMAX > 256)
        umap_level_regs<8>(h, level, mk, nk, n2, cur, nxt, bmin, bar, tid);
      uint16_t* sw = cur;
      cur = nxt;
      nxt = sw;
      nk = mk;
      if (mk == n) break;
      continue;
    }
    for (uint32_t t = tid; t < mk; t += NT) {
      const uint32_t e = t < nk ? cur[t] : t;
      atomicMin(&bmin[umap_bucket(h[e], level)], t);
    }
    bar();
    for (uint32_t t = tid; t < n2; t += NT) {
      uint32_t key = 0xffffffffu;
      if (t < mk) {
        const uint32_t e = t < nk ? cur[t] : t;
        key = (0xffffu - bmin[umap_bucket(h[e], level)]) << 16 | (0xffffu - t);
      }
      keys[t] = key;
    }
    for (uint32_t k = 2; k <= n2; k <<= 1)
      for (uint32_t j = k >> 1; j > 0; j >>= 1) {
        bar();
        for (uint32_t i = tid; i < n2; i += NT) {
          const uint32_t ixj = i ^ j;
          if (ixj > i) {
            const uint32_t x = keys[i], y = keys[ixj];
            if ((x > y) == ((i & k) == 0)) {
              keys[i] = y;
              keys[ixj] = x;
            }
          }
        }
      }
    bar();
    for (uint32_t j = tid; j < mk; j += NT) {
      const uint32_t t = 0xffffu - (keys[j] & 0xffffu);
      nxt[j] = (uint16_t)(t < nk ? cur[t] : t);
    }
    bar();
    uint16_t* sw = cur;
    cur = nxt;
    nxt = sw;
    nk = mk;
    if (mk == n) break;
  }
  for (uint32_t j = tid; j < n; j += NT) order[j] = (OutT)cur[j];
}

// ---- per-frame region tables -----------------------------------------------------------------------
constexpr uint32_t KPM_KMAX = 4096;     // entries (and code groups) per REGION the general build instance holds
constexpr uint32_t KPM_TABLE_MAX = 16384;  // descriptor-table stride (kmax) supported by the matcher
constexpr uint32_t KPM_BUCKETS = 5087;  // bucket scratch (covers 4096 groups)
constexpr uint32_t HDR_U32 = 64;        // 8 regions x 8 words

// header words per region
enum { H_E = 0, H_G = 1, H_POFF = 2, H_GOFF = 3, H_HTOFF = 4, H_HTMASK = 5, H_W2 = 6, H_W1 = 7 };

struct KpmWorkspace {
  uint32_t* pool = nullptr;  // (n_cap + 1) frames x stride_u32
  uint32_t* tickets = nullptr;  // n_cap pairs x 8 regions x 3 x 2 (key, count)
  uint32_t* needs_large = nullptr;  // n_cap + 1 flags written by kpm_build_small
  uint32_t* work = nullptr;         // 1 + n_cap * 8 work items for kpm_match<LARGE>
  uint32_t* big = nullptr;          // 1 + 24 * (n_cap + 1): items left to rmx_kpm_resolve
  uint32_t n_cap = 0, kmax = 0;
  size_t bytes = 0;
};
__host__ __device__ inline uint32_t kpm_stride_u32(uint32_t kmax) { return HDR_U32 + 48 * kmax + 136; }
__host__ __device__ inline uint32_t kpm_off_pts(uint32_t) { return HDR_U32; }
__host__ __device__ inline uint32_t kpm_off_gstart(uint32_t kmax) { return HDR_U32 + 4 * kmax; }
__host__ __device__ inline uint32_t kpm_off_gfirst(uint32_t kmax) { return HDR_U32 + 8 * kmax + 8; }
__host__ __device__ inline uint32_t kpm_off_giter(uint32_t kmax) { return HDR_U32 + 12 * kmax + 8; }
__host__ __device__ inline uint32_t kpm_off_ht(uint32_t kmax) { return HDR_U32 + 16 * kmax + 8; }
// 16-byte code of every group (by group id): saves the matcher the gfirst -> descriptor-row hops.  A code has 13
// bytes: the last dword carries byte 12 in its low byte and, above it, the low 24 bits of the code's hash
// (kpr::code_hash), which the builder has at hand — the matcher takes its probe slot from there instead of hashing
// 13 bytes again per lookup (two records with the same code carry the same bits: comparisons are not affected).
__host__ __device__ inline uint32_t kpm_off_gcode(uint32_t kmax) { return HDR_U32 + 32 * kmax + 136; }
__device__ __forceinline__ void store_gcode(uint32_t* pool, uint32_t kmax, uint32_t g, const uint32_t c[4], uint32_t hash_lo) {
  reinterpret_cast<uint4*>(pool + kpm_off_gcode(kmax))[g] = make_uint4(c[0], c[1], c[2], (c[3] & 0xffu) | hash_lo << 8);
}
__device__ __forceinline__ uint32_t gcode_hash(const uint32_t c[4]) { return c[3] >> 8; }  // (low 24 bits: masks are <= 2^14)
__device__ __forceinline__ void load_gcode(const uint32_t* pool, uint32_t kmax, uint32_t g, uint32_t c[4]) {
  const uint4 v = reinterpret_cast<const uint4*>(pool + kpm_off_gcode(kmax))[g];
  c[0] = v.x;
  c[1] = v.y;
  c[2] = v.z;
  c[3] = v.w;
}

inline size_t kpm_workspace_bytes(const KpmWorkspace& w) { return w.bytes; }
inline void kpm_workspace_free(KpmWorkspace& w) {
  if (w.pool) (void)hipFree(w.pool);
  if (w.tickets) (void)hipFree(w.tickets);
  if (w.needs_large) (void)hipFree(w.needs_large);
  if (w.work) (void)hipFree(w.work);
  if (w.big) (void)hipFree(w.big);
  w = KpmWorkspace{};
}

struct KpmBuildParams {
  const rmx_desc* desc;        // n tables, stride kmax
  const uint32_t* kcount;
  const rmx_desc* halo_desc;   // optional table of the frame before frame 0 (pool slot n)
  const uint32_t* halo_kcount;
  uint32_t* pool;
  uint32_t* status;
  uint32_t n, kmax, nreg;  // kmax: rows per frame the pool is laid out for (<= KPM_TABLE_MAX)
  uint32_t tstride;        // rows between two frames' tables in desc (>= kmax; 0 = kmax)
  // work list of the global-memory matcher (rmx_kpm_resolve): (pair * nreg + region) items whose
  // tickets the LDS instances cannot compute; null = report an overflow as a capacity error
  uint32_t* big;
};

__device__ __forceinline__ void load_code(const rmx_desc* d, uint32_t c[4]) {
  uint4 v;  // 18-byte rows, 2-byte aligned: ONE unaligned 16-byte load (seven 2-byte loads otherwise)
  __builtin_memcpy(&v, d, 16);
  c[0] = v.x;
  c[1] = v.y;
  c[2] = v.z;
  c[3] = v.w & 0xffu;
}
// x | y << 16 of a row: one unaligned 4-byte load
__device__ __forceinline__ uint32_t load_xy(const rmx_desc* d) {
  uint32_t v;
  __builtin_memcpy(&v, reinterpret_cast<const uint8_t*>(d) + 14, 4);
  return v;
}
__device__ __forceinline__ bool same_code(const uint32_t a[4], const uint32_t b[4]) {
  return a[0] == b[0] && a[1] == b[1] && a[2] == b[2] && a[3] == b[3];
}

// Exclusive prefix sum of one value per thread over an NT-thread block (LDS scratch: 8 u32).
template <uint32_t NT>
__device__ inline uint32_t block_exscan(uint32_t v, uint32_t* scratch, uint32_t* total) {
  const uint32_t lane = threadIdx.x & 63u, wv = threadIdx.x >> 6;
  uint32_t incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t t = (uint32_t)__shfl_up((int)incl, o);
    if (lane >= (uint32_t)o) incl += t;
  }
  if (NT == 64) {
    *total = (uint32_t)__shfl((int)incl, 63);
    return incl - v;
  }
  __syncthreads();
  if (lane == 63) scratch[wv] = incl;
  __syncthreads();
  uint32_t base = 0, tot = 0;
  for (uint32_t w = 0; w < NT / 64; ++w) {
    uint32_t s = scratch[w];
    if (w < wv) base += s;
    tot += s;
  }
  *total = tot;
  return base + incl - v;
}

template <uint32_t NT>
__device__ inline void bitonic_sort_u32(uint32_t* a, uint32_t n2) {
  for (uint32_t k = 2; k <= n2; k <<= 1) {
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      __syncthreads();
      for (uint32_t i = threadIdx.x; i < n2; i += NT) {
        uint32_t ixj = i ^ j;
        if (ixj > i) {
          uint32_t x = a[i], y = a[ixj];
          bool up = (i & k) == 0;
          if ((x > y) == up) {
            a[i] = y;
            a[ixj] = x;
          }
        }
      }
    }
  }
  __syncthreads();
}

// General instance (one workgroup per frame, regions one after the other).  It runs after
// kpm_build_small and takes only the frames that kernel marked in `needs_large`.
template <uint32_t KC, uint32_t NT, uint32_t BUCKETS>
__device__ __forceinline__ void kpm_build_frame(const KpmBuildParams& p, const uint32_t* needs_large, uint32_t fslot) {
  __shared__ uint16_t ent[KC];       // scan indices of the region's entries, scan order
  __shared__ uint16_t slot_of[KC];   // hash slot of entry i
  __shared__ uint16_t gid_of[KC];    // rank of a group's first entry -> group id
  __shared__ uint32_t htl[2 * KC];   // slot -> smallest entry rank with that code
  __shared__ uint32_t keys[KC];
  __shared__ uint64_t hg[KC];        // the current region's group hashes in insertion order
  __shared__ uint16_t seq_a[KC], seq_b[KC];
  __shared__ uint32_t bmin[BUCKETS];
  __shared__ uint32_t scan_scratch[8];
  __shared__ uint32_t s_w2, s_w1;

  // fslot: 0..n-1 frames of the batch, n = halo
  const bool is_halo = fslot == p.n;
  const rmx_desc* desc = is_halo ? p.halo_desc : p.desc + (size_t)fslot * (p.tstride ? p.tstride : p.kmax);
  uint32_t k = is_halo ? p.halo_kcount[0] : p.kcount[fslot];
  if (k > p.kmax) k = p.kmax;
  if (needs_large && needs_large[fslot] != 3u) return;  // built by the wave-per-region kernels (uniform)
  uint32_t* pool = p.pool + (size_t)fslot * kpm_stride_u32(p.kmax);
  uint32_t* hdr = pool;
  uint32_t* pts = pool + kpm_off_pts(p.kmax);
  uint32_t* gstart = pool + kpm_off_gstart(p.kmax);
  uint32_t* gfirst = pool + kpm_off_gfirst(p.kmax);
  uint32_t* ght = pool + kpm_off_ht(p.kmax);
  const uint32_t tid = threadIdx.x;
  const uint32_t EMPTY = 0xffffffffu;

  uint32_t p_off = 0, g_off = 0, ht_off = 0;  // running pool offsets (uniform)
  const uint32_t seg = (k + NT - 1) / NT;  // contiguous entries per thread

  for (uint32_t r = 0; r < p.nreg; ++r) {
    // 1. compact the region's entries, preserving scan order
    uint32_t cnt = 0;
    const uint32_t i0 = tid * seg, i1 = min(k, i0 + seg);
    for (uint32_t i = i0; i < i1; ++i) cnt += (desc[i].region_mask >> r) & 1u;
    uint32_t e = 0;
    uint32_t w = block_exscan<NT>(cnt, scan_scratch, &e);
    if (e > KC) {  // more entries in one region than this instance holds: region left empty here
      if (tid == 0) {
        if (p.big) {  // ... and the tickets of the two pairs it takes part in left to rmx_kpm_resolve
          const uint32_t q0 = is_halo ? 0u : fslot;
          const bool two = !is_halo && fslot + 1 < p.n;
          const uint32_t at = atomicAdd(&p.big[0], two ? 2u : 1u);
          p.big[1 + at] = q0 * p.nreg + r;
          if (two) p.big[2 + at] = (fslot + 1) * p.nreg + r;
          atomicOr(&p.status[1], 128u);
        } else {
          atomicOr(&p.status[1], 64u);
        }
      }
      e = 0;
      w = 0;
      cnt = 0;
    }
    for (uint32_t i = i0; i < i1 && cnt; ++i)
      if ((desc[i].region_mask >> r) & 1u) ent[w++] = (uint16_t)i;
    uint32_t slots = 16;
    while (slots < 2 * e) slots <<= 1;
    const uint32_t mask = slots - 1;
    for (uint32_t s = tid; s < slots; s += NT) htl[s] = EMPTY;
    if (tid == 0) {
      s_w2 = 0;
      s_w1 = 0;
    }
    __syncthreads();
    // 2. group by code: slot value = smallest entry rank holding that code
    for (uint32_t i = tid; i < e; i += NT) {
      uint32_t c[4];
      load_code(desc + ent[i], c);
      uint32_t wgt = c[3] & 0xfu;
      if (wgt == 2) atomicAdd(&s_w2, 1u);
      if (wgt == 1) atomicAdd(&s_w1, 1u);
      uint32_t s = fnv_code_lo(c) & mask;
      for (;;) {
        uint32_t v = atomicCAS(&htl[s], EMPTY, i);
        if (v == EMPTY) break;
        uint32_t c2[4];
        load_code(desc + ent[v], c2);
        if (same_code(c, c2)) {
          atomicMin(&htl[s], i);
          break;
        }
        s = (s + 1) & mask;
      }
      slot_of[i] = (uint16_t)s;
    }
    __syncthreads();
    // 3. order entries by (first occurrence of their code, scan order)
    uint32_t n2 = 1;
    while (n2 < e) n2 <<= 1;
    for (uint32_t i = tid; i < n2; i += NT) keys[i] = i < e ? (htl[slot_of[i]] << 16 | i) : 0xffffffffu;
    bitonic_sort_u32<NT>(keys, n2);
    // 4. groups: ids in first-occurrence order; points, starts, first indices, hashes
    const uint32_t eseg = (e + NT - 1) / NT;
    const uint32_t j0 = tid * eseg, j1 = min(e, j0 + eseg);
    uint32_t starts = 0;
    for (uint32_t j = j0; j < j1; ++j) starts += (j == 0 || (keys[j] >> 16) != (keys[j - 1] >> 16));
    uint32_t G = 0;
    uint32_t gid = block_exscan<NT>(starts, scan_scratch, &G);
    for (uint32_t j = j0; j < j1; ++j) {
      uint32_t first = keys[j] >> 16, i = keys[j] & 0xffffu;
      const rmx_desc* d = desc + ent[i];
      pts[p_off + j] = load_xy(d);
      if (j == 0 || first != (keys[j - 1] >> 16)) {
        gstart[g_off + r + gid] = p_off + j;
        gfirst[g_off + gid] = ent[first];
        gid_of[first] = (uint16_t)gid;
        uint32_t c[4];
        load_code(desc + ent[first], c);
        hg[gid] = fnv_code(c);
        store_gcode(pool, p.kmax, g_off + gid, c, (uint32_t)hg[gid]);
        ++gid;
      }
    }
    __syncthreads();
    // 5. export the code index: slot -> group id + 1
    for (uint32_t s = tid; s < slots; s += NT) {
      uint32_t v = htl[s];
      ght[ht_off + s] = v == EMPTY ? 0u : (uint32_t)gid_of[v] + 1u;
    }
    if (tid == 0) {
      gstart[g_off + r + G] = p_off + e;  // sentinel (one extra slot per region)
      uint32_t* h = hdr + r * 8;
      h[H_E] = e;
      h[H_G] = G;
      h[H_POFF] = p_off;
      h[H_GOFF] = g_off;
      h[H_HTOFF] = ht_off;
      h[H_HTMASK] = mask;
      h[H_W2] = s_w2;
      h[H_W1] = s_w1;
    }
    __syncthreads();
    // 6. iteration order of this region's code map (keys[] is free again)
    umap_order_par<NT>(hg, G, keys, seq_a, seq_b, bmin, BlockBarrier{}, pool + kpm_off_giter(p.kmax) + g_off, tid);
    p_off += e;
    g_off += G;
    ht_off += slots;
    __syncthreads();
  }
}

// ---- small frames: one wave per region ---------------------------------------------------------------
// Frames with at most 256 keypoints and at most 64 entries in every region (the common case:
// the synthetic streams have ~10 entries per region) are built by kpm_build_small: eight
// waves, wave r owns region r, everything wave-synchronous (ballot compaction, a 64-key
// bitonic sort in registers, one lane replaying libstdc++'s node order), two block barriers
// to exchange the regions' sizes.  It runs before kpm_build<2048,...>, which then takes
// only the frames marked in `needs_large`.
// The kernel is VALU-issue-bound (~650 wave-instructions per region, 52 M per 10 000 sparse
// frames = the 94 us it took with a wave per region), and a sparse region fills 10 of its 64
// lanes: the first pass gives a region HALF a wave (LPR = 32 lanes, two regions per wave), frames
// with a region of 33-64 entries go on to the wave-per-region instance (LPR = 64, tier 4).

// Iteration order of a code map of n <= 32 groups, for the two 32-lane halves of a wave side by
// side (the halves differ in n: control flow stays wave-uniform, finished halves idle).  Same
// levels, keys and result as umap_order_par.
template <typename OutT>
__device__ __forceinline__ void umap_order_half(const uint64_t* h, uint32_t n, uint16_t* seq_a, uint16_t* seq_b, uint32_t* bmin,
                                                OutT* order, uint32_t l) {
  const WaveBarrier bar{};
  uint16_t* cur = seq_a;
  uint16_t* nxt = seq_b;
  uint32_t nk = 0;
#pragma unroll
  for (int level = 0; level < 3; ++level) {
    const uint32_t B = level == 0 ? 13u : level == 1 ? 29u : 59u, mk = min(B, n);
    const bool on = level == 0 || nk < n;
    if (__builtin_amdgcn_ballot_w64(on) == 0) break;  // uniform
    if (on)
      for (uint32_t b = l; b < B; b += 32) bmin[b] = 0xffffffffu;
    bar();
    uint32_t bkt = 0;
    const bool live = on && l < mk;
    if (live) {
      const uint32_t e = l < nk ? cur[l] : l;
      bkt = umap_bucket(h[e], level);
      atomicMin(&bmin[bkt], l);
    }
    bar();
    uint32_t key = live ? (0xffffu - bmin[bkt]) << 16 | (0xffffu - l) : 0xffffffffu;
    key = wave_sort1(key, level == 0 ? 16u : 32u, l);
    if (live) {
      const uint32_t t = 0xffffu - (key & 0xffffu);
      nxt[l] = (uint16_t)(t < nk ? cur[t] : t);
    }
    bar();
    if (on) {
      uint16_t* sw = cur;
      cur = nxt;
      nxt = sw;
      nk = mk;
    }
  }
  if (l < n) order[l] = (OutT)cur[l];
}

// LPR lanes per region; FIRST: the pass over every frame (writes every tier word), else the
// frames of tier 4.  Returns through the tier word: 0 built, 1 / 3 / 4 = the instance to go to.
template <uint32_t LPR, bool FIRST>
__device__ __forceinline__ void kpm_build_small_frame(const KpmBuildParams& p, uint32_t* needs_large, uint32_t fslot) {
  static_assert(LPR == 32 || LPR == 64, "half a wave or a wave per region");
  constexpr uint32_t KS_RC = LPR;                     // entries per region
  constexpr uint32_t KS_BK = LPR == 64 ? 127u : 59u;  // buckets of the last level a map of KS_RC groups reaches
  __shared__ uint16_t ent[8][KS_RC];
  __shared__ uint32_t htl[8][2 * KS_RC];
  __shared__ uint64_t hg[8][KS_RC];
  __shared__ uint16_t seq_a[8][KS_RC], seq_b[8][KS_RC], gid_of[8][KS_RC];
  __shared__ uint32_t okeys[8][KS_RC], bmin[8][KS_BK];
  __shared__ uint32_t s_e[8], s_g[8];
  __shared__ uint32_t s_big, s_maxe;
  // the frame's rows (this instance: k <= 256), staged once and coalesced: region masks, codes and coordinates
  // come from LDS afterwards instead of five dependent trips to global memory per frame
  __shared__ uint4 s_row[256];   // code bytes 0..12, region mask, x
  __shared__ uint32_t s_xy[256];  // x | y << 16

  const bool is_halo = fslot == p.n;
  const rmx_desc* desc = is_halo ? p.halo_desc : p.desc + (size_t)fslot * (p.tstride ? p.tstride : p.kmax);
  uint32_t k = is_halo ? p.halo_kcount[0] : p.kcount[fslot];
  if (FIRST && k > p.kmax && p.big) {
    // more keypoints than the pool is laid out for (a frame that is mostly noise): all eight
    // regions are left empty here; tier 5 = kpm_build hands the tickets of the frame's two pairs
    // to rmx_kpm_resolve, whose tables have no size limit (not from this kernel: its first
    // workgroup is still clearing the lists' counters)
    if (threadIdx.x < 8 * 8) (p.pool + (size_t)fslot * kpm_stride_u32(p.kmax))[threadIdx.x] = 0;
    if (threadIdx.x == 0) needs_large[fslot] = 5u;
    return;
  }
  if (k > p.kmax) k = p.kmax;
  const uint32_t lane = threadIdx.x & (LPR - 1u);  // lane inside the region's group
  const uint32_t r = LPR == 64 ? (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) : threadIdx.x / LPR;
  const uint32_t gshift = LPR == 64 ? 0u : (threadIdx.x & 32u);  // my group's bits inside a wave ballot
  auto gballot = [&](bool pred) -> uint64_t {
    const uint64_t b = __builtin_amdgcn_ballot_w64(pred);
    return LPR == 64 ? b : (b >> gshift) & 0xffffffffull;
  };
  auto gcount_below = [&](uint64_t m) -> uint32_t { return (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull)); };
  const uint32_t EMPTY = 0xffffffffu;
  if (threadIdx.x == 0) {
    s_big = k > 256 ? 1u : 0u;
    s_maxe = 0;
  }
  if (k <= 256 && threadIdx.x < k) {
    uint4 v;
    __builtin_memcpy(&v, desc + threadIdx.x, 16);
    s_row[threadIdx.x] = v;
    s_xy[threadIdx.x] = load_xy(desc + threadIdx.x);
  }
  auto code_of = [&](uint32_t i, uint32_t (&cc)[4]) {
    const uint4 v = s_row[i];
    cc[0] = v.x, cc[1] = v.y, cc[2] = v.z, cc[3] = v.w & 0xffu;
  };
  __syncthreads();
  const bool mine = r < p.nreg && k <= 256;
  // 1. region entries in scan order
  uint32_t e = 0;
  if (k <= 256) {  // (uniform; every lane takes part in the ballots)
    for (uint32_t base = 0; base < k; base += LPR) {
      uint32_t i = base + lane;
      bool in = mine && i < k && ((s_row[i].w >> (8 + r)) & 1u);
      uint64_t bal = gballot(in);
      uint32_t pos = e + gcount_below(bal);
      if (in && pos < KS_RC) ent[r][pos] = (uint16_t)i;
      e += (uint32_t)__builtin_popcountll(bal);
    }
    if (e > KS_RC && lane == 0) {
      atomicOr(&s_big, 1u);
      atomicMax(&s_maxe, e);
    }
  }
  __syncthreads();
  if (s_big) {  // uniform for the block
    // tier: wave-per-region instances (4: <= 64 entries per region, 1, 2) or general (3)
    if (threadIdx.x == 0) needs_large[fslot] = k > 4096 ? 3u : (LPR == 32 && k <= 256 && s_maxe <= 64) ? 4u : 1u;
    return;
  }
  if (threadIdx.x == 0) needs_large[fslot] = 0u;
  // 2. group by code (slot value = smallest entry rank with that code), weights
  uint32_t c[4] = {0, 0, 0, 0}, myslot = 0, w2 = 0, w1 = 0, slots = 16, G = 0, gid = 0, first = 0;
  uint64_t h64 = 0;
  bool start = false;
  uint32_t key = 0xffffffffu;
  {
    while (slots < 2 * e) slots <<= 1;
    const uint32_t mask = slots - 1;
    if (mine)
      for (uint32_t s = lane; s < slots; s += LPR) htl[r][s] = EMPTY;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    if (mine && lane < e) {
      code_of(ent[r][lane], c);
      h64 = fnv_code(c);  // (once per entry: its low bits are the slot, a group's first entry hands it to the group)
      uint32_t s = (uint32_t)h64 & mask;
      for (;;) {
        uint32_t v = atomicCAS(&htl[r][s], EMPTY, lane);
        if (v == EMPTY) break;
        uint32_t c2[4];
        code_of(ent[r][v], c2);
        if (same_code(c, c2)) {
          atomicMin(&htl[r][s], lane);
          break;
        }
        s = (s + 1) & mask;
      }
      myslot = s;
    }
    w2 = (uint32_t)__builtin_popcountll(gballot(mine && lane < e && (c[3] & 0xfu) == 2u));
    w1 = (uint32_t)__builtin_popcountll(gballot(mine && lane < e && (c[3] & 0xfu) == 1u));
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    // 3. order entries by (first occurrence of their code, scan order): bitonic sort of one key per lane
    if (mine && lane < e) key = htl[r][myslot] << 16 | lane;
    // (unused lanes hold 0xffffffff; a region of the sparse stream has ~10 entries: 16 lanes, 10 steps;
    // the size is the wave's: both halves run the same network)
    const uint32_t n2 = __builtin_amdgcn_ballot_w64(e > 16) == 0 ? 16u : LPR;
    key = wave_sort1(key, n2, lane);
    // 4. groups in first-occurrence order
    const uint32_t prev = (uint32_t)__shfl_up((int)key, 1);
    start = mine && lane < e && (lane == 0 || (key >> 16) != (prev >> 16));
    const uint64_t sb = gballot(start);
    G = (uint32_t)__builtin_popcountll(sb);
    gid = gcount_below(sb);  // group id of a start lane
    first = key >> 16;
    if (lane == 0 && r < 8) {
      s_e[r] = mine ? e : 0u;
      s_g[r] = mine ? G : 0u;
    }
  }
  __syncthreads();
  // pool offsets: exclusive sums over the regions before this one
  uint32_t p_off = 0, g_off = 0, ht_off = 0;
  for (uint32_t q = 0; q < r; ++q) {
    p_off += s_e[q];
    g_off += s_g[q];
    uint32_t sl = 16;
    while (sl < 2 * s_e[q]) sl <<= 1;
    ht_off += sl;
  }
  uint32_t* pool = p.pool + (size_t)fslot * kpm_stride_u32(p.kmax);
  uint32_t* pts = pool + kpm_off_pts(p.kmax);
  uint32_t* gstart = pool + kpm_off_gstart(p.kmax);
  uint32_t* gfirst = pool + kpm_off_gfirst(p.kmax);
  uint32_t* ght = pool + kpm_off_ht(p.kmax);
  // the hash of a group's first entry sits in the lane with that entry's rank (same half-wave)
  const uint32_t src = (threadIdx.x & 63u & ~(LPR - 1u)) + (first & (LPR - 1u));
  const uint64_t hf = (uint64_t)(uint32_t)__shfl((int)(uint32_t)h64, (int)src) |
                      (uint64_t)(uint32_t)__shfl((int)(uint32_t)(h64 >> 32), (int)src) << 32;
  if (mine && lane < e) {
    pts[p_off + lane] = s_xy[ent[r][key & 0xffffu]];
    if (start) {
      gstart[g_off + r + gid] = p_off + lane;
      gfirst[g_off + gid] = ent[r][first];
      uint32_t cf[4];
      code_of(ent[r][first], cf);
      hg[r][gid] = hf;
      store_gcode(pool, p.kmax, g_off + gid, cf, (uint32_t)hf);
      gid_of[r][first] = (uint16_t)gid;  // rank of a group's first entry -> group id
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
  // export the code index: slot -> group id + 1
  if (mine)
    for (uint32_t sidx = lane; sidx < slots; sidx += LPR) {
      uint32_t v = htl[r][sidx];
      ght[ht_off + sidx] = v == EMPTY ? 0u : (uint32_t)gid_of[r][v] + 1u;
    }
  if (mine && lane == 0) {
    gstart[g_off + r + G] = p_off + e;
    uint32_t* h = pool + r * 8;
    h[H_E] = e;
    h[H_G] = G;
    h[H_POFF] = p_off;
    h[H_GOFF] = g_off;
    h[H_HTOFF] = ht_off;
    h[H_HTMASK] = slots - 1;
    h[H_W2] = w2;
    h[H_W1] = w1;
  }
  // iteration order of this region's code map, wave-parallel
  if constexpr (LPR == 64) {
    if (!mine) return;  // (wave-uniform)
    umap_order_par<64, 64>(hg[r], G, okeys[r], seq_a[r], seq_b[r], bmin[r], WaveBarrier{},
                           pool + kpm_off_giter(p.kmax) + g_off, lane);
  } else {
    umap_order_half(hg[r & 7u], mine ? G : 0u, seq_a[r & 7u], seq_b[r & 7u], bmin[r & 7u], pool + kpm_off_giter(p.kmax) + g_off, lane);
  }
}

// first pass, every frame: half a wave per region
__global__ __launch_bounds__(256) void kpm_build_small(KpmBuildParams p, uint32_t* needs_large, uint32_t* work, uint32_t work_stride) {
  if (blockIdx.x == 0 && threadIdx.x == 0) {  // counters of the matcher's work lists (used by later launches)
    work[0] = 0;
    work[work_stride] = 0;
    work[2 * work_stride] = 0;
    if (p.big) p.big[0] = 0;
    p.status[13] = 0;
    p.status[14] = 0;
  }
  kpm_build_small_frame<32, true>(p, needs_large, blockIdx.x);
}

// ---- medium frames: one wave per region, up to 256 entries per region ---------------------------------
// Same division of labour as kpm_build_small (wave r owns region r, no block barriers inside
// the region work) for the frames that kernel passed on (tier 1): a region's entries are
// walked 64 at a time, the sort runs in the wave's LDS slice.  Frames with a bigger region
// go on to the general instance (tier 2).

__device__ __forceinline__ uint32_t wave_exscan(uint32_t v, uint32_t lane, uint32_t* total) {
  uint32_t incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    uint32_t t = (uint32_t)__shfl_up((int)incl, o);
    if (lane >= (uint32_t)o) incl += t;
  }
  *total = (uint32_t)__shfl((int)incl, 63);
  return incl - v;
}

// RC = entries per region the instance holds (256: 53 + 25 KB of LDS, two workgroups per CU;
// 512: 104 + 40 KB, one); a frame enters at tier TIN and leaves at 0 (built) or TIN + 1.
template <uint32_t RC, uint32_t TIN>
__device__ __forceinline__ void kpm_build_medium_frame(const KpmBuildParams& p, uint32_t* tier, uint32_t fslot) {
  // (the map-order scratch lives in arrays that are dead by then: see the umap_order_par call;
  // 53 KB instead of 70 for RC = 256 lets a CU hold three workgroups instead of two)
  __shared__ uint16_t ent[8][RC], slot_of[8][RC], gid_of[8][RC];
  __shared__ uint32_t htl[8][2 * RC], keys[8][RC];
  __shared__ uint64_t hg[8][RC];
  static_assert(RC / 2 + (RC == 256 ? 257 : 541) <= 2 * RC, "seq_a + bmin must fit the region's hash table");
  __shared__ uint32_t s_e[8], s_g[8];
  __shared__ uint32_t s_big;
  // The frame's table rows, staged once by the whole workgroup (coalesced): every later phase reads codes,
  // region masks and coordinates from LDS; a frame with more rows than fit goes on to the next instance.  (From global memory each of the eight region waves
  // scanned all k masks, and grouping / numbering / export were chains of dependent 16-byte loads: 73 us per
  // frame for ~1 100 rows, four fifths of it waiting for those loads — tools/kpm_phases.py.)
  constexpr uint32_t KS = RC == 256 ? 1280 : 2048;
  __shared__ uint4 s_row[KS];   // code bytes 0..12, region mask, x
  __shared__ uint32_t s_xy[KS];  // x | y << 16

  if (tier[fslot] != TIN) return;  // uniform: built already, or not yet examined
#if defined(RMX_KPM_STAMP)  /* diagnostic build (tools/kpm_phases.py): time per phase, summed over the region waves, in status[40..] */
  uint64_t stamp_t = __builtin_amdgcn_s_memtime();
  uint32_t stamp_d[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, stamp_i = 0;  // (added to the status words at the end: nothing of it inside the phases)
#define KPM_STAMP()                                            \
  {                                                            \
    const uint64_t now = __builtin_amdgcn_s_memtime();         \
    stamp_d[stamp_i++] = (uint32_t)(now - stamp_t);            \
    stamp_t = now;                                             \
  }
#else
#define KPM_STAMP()
#endif
  const bool is_halo = fslot == p.n;
  const rmx_desc* desc = is_halo ? p.halo_desc : p.desc + (size_t)fslot * (p.tstride ? p.tstride : p.kmax);
  uint32_t k = is_halo ? p.halo_kcount[0] : p.kcount[fslot];
  if (k > p.kmax) k = p.kmax;
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t r = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const uint32_t EMPTY = 0xffffffffu;
  const WaveBarrier wbar{};
  if (k > KS) {  // uniform: more rows than the staging area holds (no region-sized instance reads rows from global memory:
                 // a global load in a loop that also stores makes every store wait for the one before, s_waitcnt vmcnt(0))
    if (threadIdx.x == 0) tier[fslot] = TIN + 1u;
    return;
  }
  if (threadIdx.x == 0) s_big = 0u;
  for (uint32_t i = threadIdx.x; i < k; i += blockDim.x) {
    uint4 v;
    __builtin_memcpy(&v, desc + i, 16);
    s_row[i] = v;
    s_xy[i] = load_xy(desc + i);
  }
  auto code_of = [&](uint32_t i, uint32_t (&c)[4]) {
    const uint4 v = s_row[i];
    c[0] = v.x, c[1] = v.y, c[2] = v.z, c[3] = v.w & 0xffu;
  };
  __syncthreads();
  const bool mine = r < p.nreg;
  // 1. region entries in scan order
  uint32_t e = 0;
  if (mine) {
    for (uint32_t base = 0; base < k; base += 64) {
      uint32_t i = base + lane;
      bool in = i < k && ((s_row[i].w >> (8 + r)) & 1u);
      uint64_t bal = __builtin_amdgcn_ballot_w64(in);
      uint32_t pos = e + mbcnt(bal);
      if (in && pos < RC) ent[r][pos] = (uint16_t)i;
      e += (uint32_t)__builtin_popcountll(bal);
    }
    if (e > RC && lane == 0) atomicOr(&s_big, 1u);
  }
  __syncthreads();
  if (s_big) {
    if (threadIdx.x == 0) tier[fslot] = TIN + 1u;
    return;
  }
  if (threadIdx.x == 0) tier[fslot] = 0u;
  KPM_STAMP();  // 40: entries
  uint32_t slots = 16, G = 0, w2 = 0, w1 = 0;
  if (mine) {
    // 2. group by code
    while (slots < 2 * e) slots <<= 1;
    const uint32_t mask = slots - 1;
    for (uint32_t s = lane; s < slots; s += 64) htl[r][s] = EMPTY;
    wbar();
    for (uint32_t i = lane; i < e; i += 64) {
      uint32_t c[4];
      code_of(ent[r][i], c);
      const uint32_t wgt = c[3] & 0xfu;
      w2 += wgt == 2u;
      w1 += wgt == 1u;
      uint32_t s = fnv_code_lo(c) & mask;
      for (;;) {
        uint32_t v = atomicCAS(&htl[r][s], EMPTY, i);
        if (v == EMPTY) break;
        uint32_t c2[4];
        code_of(ent[r][v], c2);
        if (same_code(c, c2)) {
          atomicMin(&htl[r][s], i);
          break;
        }
        s = (s + 1) & mask;
      }
      slot_of[r][i] = (uint16_t)s;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      w2 += (uint32_t)__shfl_xor((int)w2, o);
      w1 += (uint32_t)__shfl_xor((int)w1, o);
    }
    wbar();
    KPM_STAMP();  // 41: grouped
    // 3. order entries by (first occurrence of their code, scan order)
    uint32_t n2 = 1;
    while (n2 < e) n2 <<= 1;
    {
      uint32_t kreg[RC / 64];
#pragma unroll
      for (uint32_t sl = 0; sl < RC / 64; ++sl) {
        const uint32_t i = sl * 64 + lane;
        kreg[sl] = i < e ? (htl[r][slot_of[r][i]] << 16 | i) : 0xffffffffu;
      }
      wave_sort<RC / 64>(kreg, n2, lane);
#pragma unroll
      for (uint32_t sl = 0; sl < RC / 64; ++sl) keys[r][sl * 64 + lane] = kreg[sl];
    }
    wbar();
    KPM_STAMP();  // 42: sorted
    // 4. number the groups (first-occurrence order)
    for (uint32_t base = 0; base < e; base += 64) {
      const uint32_t j = base + lane;
      const bool st = j < e && (j == 0 || (keys[r][j] >> 16) != (keys[r][j - 1] >> 16));
      const uint64_t sb = __builtin_amdgcn_ballot_w64(st);
      if (st) {
        const uint32_t gid = G + mbcnt(sb), first = keys[r][j] >> 16;
        gid_of[r][first] = (uint16_t)gid;
        uint32_t cf[4];
        code_of(ent[r][first], cf);
        hg[r][gid] = fnv_code(cf);
      }
      G += (uint32_t)__builtin_popcountll(sb);
    }
    if (lane == 0) {
      s_e[r] = e;
      s_g[r] = G;
    }
  } else if (lane == 0 && r < 8) {
    s_e[r] = 0;
    s_g[r] = 0;
  }
  KPM_STAMP();  // 43: numbered
  __syncthreads();
  if (!mine) return;
  KPM_STAMP();  // 44: waited for the other regions
  uint32_t p_off = 0, g_off = 0, ht_off = 0;
  for (uint32_t q = 0; q < r; ++q) {
    p_off += s_e[q];
    g_off += s_g[q];
    uint32_t sl = 16;
    while (sl < 2 * s_e[q]) sl <<= 1;
    ht_off += sl;
  }
  uint32_t* pool = p.pool + (size_t)fslot * kpm_stride_u32(p.kmax);
  uint32_t* pts = pool + kpm_off_pts(p.kmax);
  uint32_t* gstart = pool + kpm_off_gstart(p.kmax);
  uint32_t* gfirst = pool + kpm_off_gfirst(p.kmax);
  uint32_t* ght = pool + kpm_off_ht(p.kmax);
  for (uint32_t j = lane; j < e; j += 64) {
    const uint32_t first = keys[r][j] >> 16, i = keys[r][j] & 0xffffu;
    const uint32_t row = ent[r][i];
    pts[p_off + j] = s_xy[row];
    if (j == 0 || first != (keys[r][j - 1] >> 16)) {
      const uint32_t gid = gid_of[r][first];
      gstart[g_off + r + gid] = p_off + j;
      gfirst[g_off + gid] = ent[r][first];
      uint32_t cf[4];
      code_of(row, cf);  // j is the group's first entry
      store_gcode(pool, p.kmax, g_off + gid, cf, (uint32_t)hg[r][gid]);
    }
  }
  KPM_STAMP();  // 45: points and groups exported
  for (uint32_t sidx = lane; sidx < slots; sidx += 64) {
    const uint32_t v = htl[r][sidx];
    ght[ht_off + sidx] = v == EMPTY ? 0u : (uint32_t)gid_of[r][v] + 1u;
  }
  KPM_STAMP();  // 46: code index exported
  if (lane == 0) {
    gstart[g_off + r + G] = p_off + e;
    uint32_t* h = pool + r * 8;
    h[H_E] = e;
    h[H_G] = G;
    h[H_POFF] = p_off;
    h[H_GOFF] = g_off;
    h[H_HTOFF] = ht_off;
    h[H_HTMASK] = slots - 1;
    h[H_W2] = w2;
    h[H_W1] = w1;
  }
  wbar();
  KPM_STAMP();  // 47: header
  // iteration order of this region's code map.  Scratch: keys[r] is free again; the region's
  // hash table (exported above) now holds seq_a (RC u16) and the bucket minima behind it, and
  // slot_of[r] (last read when the sort keys were formed) holds seq_b
  umap_order_par<64, RC>(hg[r], G, keys[r], reinterpret_cast<uint16_t*>(htl[r]), slot_of[r], htl[r] + RC / 2, wbar,
                         pool + kpm_off_giter(p.kmax) + g_off, lane);
  KPM_STAMP();  // 48: map order
#if defined(RMX_KPM_STAMP)
  if ((threadIdx.x & 63u) == 0)
    for (uint32_t i = 0; i < 9; ++i) atomicAdd(&p.status[40 + i], stamp_d[i]);
#endif
#undef KPM_STAMP
}

// The instances behind kpm_build_small are launched on a FIXED grid; a workgroup looks at the
// tier words of all its frames side by side first and only walks them if one asks for this
// instance: on a stream where kpm_build_small builds
// everything these launches cost their launch overhead, not n mostly empty workgroups.
// status[13]: the instances of the chain that found work in this step (bit per instance, KPM_TIER_*); status[14]:
// instances a frame or a work item was waiting for but which were not launched (kpm_declare) — rmx_kpm_seq with
// the adaptive launch set, api_kpm_fgm.inc.
enum : uint32_t { KPM_TIER_SMALL64 = 1, KPM_TIER_MEDIUM256 = 2, KPM_TIER_MEDIUM512 = 4, KPM_TIER_GENERAL = 8,
                  KPM_TIER_MATCH0 = 16, KPM_TIER_MATCH1 = 32, KPM_TIER_MATCH2 = 64, KPM_TIER_ALL = 127 };
__device__ __forceinline__ void kpm_mark_used(uint32_t* status, uint32_t bit) {
  if (!(__atomic_load_n(&status[13], __ATOMIC_RELAXED) & bit)) atomicOr(&status[13], bit);
}
template <typename F>
__device__ __forceinline__ void kpm_for_tier(const uint32_t* tier, uint32_t want, uint32_t nslots, uint32_t* status, uint32_t bit, F&& frame) {
  // workgroup b owns frames b, b + G, b + 2G, ... (neighbouring frames cost alike: striding
  // spreads them over the workgroups)
  // (the FIRST wave looks at the tier words, 64 frames a trip; the others only meet it at the barrier: a workgroup
  // that finds nothing waiting — nearly all of them on a sparse stream — costs its other waves a dozen instructions)
  __shared__ uint32_t s_any;
  const uint32_t G = gridDim.x, b = blockIdx.x;
  const uint32_t K = b < nslots ? (nslots - b + G - 1) / G : 0;
  for (uint32_t base = 0; base < K; base += 64) {
    if (threadIdx.x < 64) {
      const uint32_t k = base + threadIdx.x;
      const bool mine = tier == nullptr || (k < K && tier[b + k * G] == want);
      const bool any = __builtin_amdgcn_ballot_w64(mine) != 0;
      if (threadIdx.x == 0) s_any = any ? 1u : 0u;
    }
    __syncthreads();
    const bool any = s_any != 0;
    __syncthreads();  // (s_any is rewritten by the next trip)
    if (!any) continue;
    const uint32_t end = min(K, base + 64);
    for (uint32_t kk = base; kk < end; ++kk) {
      const uint32_t fslot = b + kk * G;
      if (tier && tier[fslot] != want) continue;  // uniform
      if (threadIdx.x == 0) kpm_mark_used(status, bit);
      frame(fslot);
      __syncthreads();  // LDS is reused by the next frame
    }
  }
}
template <uint32_t KC, uint32_t NT, uint32_t BUCKETS>
__global__ __launch_bounds__(NT) void kpm_build(KpmBuildParams p, const uint32_t* needs_large, uint32_t nslots) {
  kpm_for_tier(needs_large, 3u, nslots, p.status, KPM_TIER_GENERAL, [&](uint32_t fslot) { kpm_build_frame<KC, NT, BUCKETS>(p, needs_large, fslot); });
  if (p.big)  // tier 5: frames beyond the pool's layout (kpm_build_small_frame)
    kpm_for_tier(needs_large, 5u, nslots, p.status, KPM_TIER_GENERAL, [&](uint32_t fslot) {
      if (threadIdx.x == 0) {
        const bool is_halo = fslot == p.n;
        const uint32_t q0 = is_halo ? 0u : fslot;
        const bool two = !is_halo && fslot + 1 < p.n;
        const uint32_t at = atomicAdd(&p.big[0], two ? 2u * p.nreg : p.nreg);
        for (uint32_t r = 0; r < p.nreg; ++r) {
          p.big[1 + at + r] = q0 * p.nreg + r;
          if (two) p.big[1 + at + p.nreg + r] = (fslot + 1) * p.nreg + r;
        }
        atomicOr(&p.status[1], 128u);
      }
    });
}
// tier 4: a wave per region, up to 64 entries per region
__global__ __launch_bounds__(512) void kpm_build_small64(KpmBuildParams p, uint32_t* tier, uint32_t nslots) {
  kpm_for_tier(tier, 4u, nslots, p.status, KPM_TIER_SMALL64, [&](uint32_t fslot) { kpm_build_small_frame<64, false>(p, tier, fslot); });
}
template <uint32_t RC, uint32_t TIN>
__global__ __launch_bounds__(512) void kpm_build_medium(KpmBuildParams p, uint32_t* tier, uint32_t nslots) {
  kpm_for_tier(tier, TIN, nslots, p.status, TIN == 1 ? KPM_TIER_MEDIUM256 : KPM_TIER_MEDIUM512, [&](uint32_t fslot) { kpm_build_medium_frame<RC, TIN>(p, tier, fslot); });
}

// ---- matching --------------------------------------------------------------------------------------

struct KpmMatchParams {
  const rmx_desc* desc;
  const rmx_desc* halo_desc;
  const uint32_t* pool;
  uint32_t* tickets;  // pairs x nreg x 3 x {key, count}
  uint32_t* work_in;   // [0] = count, [1..] = (pair * nreg + region) items passed on by the tier below
  uint32_t* work_out;  // same layout: items this tier passes on
  uint32_t* status;
  uint32_t n, kmax, nreg;
  uint32_t has_halo;
  uint32_t weight_switch;
  uint32_t* big;       // items left to rmx_kpm_resolve by the last tier (same layout)
  uint32_t tstride;    // rows between two frames' tables in desc (0 = kmax)
};

__device__ __forceinline__ uint32_t off_key(int32_t dx, int32_t dy) {
  return (uint32_t)(dx + 32768) << 16 | (uint32_t)(dy + 32768);
}

__device__ __forceinline__ uint64_t wave_max_u64(uint64_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    uint64_t t = (uint64_t)__shfl_xor((long long)v, o);
    v = t > v ? t : v;
  }
  return v;
}

// top_offsets(total, 3) with ties (kpm.hpp:127-159): by count, ties by iteration order of the
// totalizator map.  Insertion order of the totalizator = order of first touch among the
// pairs (hs).  One wave; scratch sized for the distinct offsets the table can hold.
template <uint32_t HSLOTS>
__device__ inline void exact_top3(const uint32_t* hk, const uint32_t* hc, const uint32_t* hs, uint64_t* skey, uint64_t* hh,
                                  uint32_t* okeys, uint16_t* seq_a, uint16_t* seq_b, uint16_t* pos_of, uint32_t* bmin,
                                  uint32_t* ticket, uint32_t lane) {
  // exact path: insertion order of the totalizator = order of first occurrence among pairs
  // compact distinct offsets
  uint32_t nd = 0;
  for (uint32_t i = 0; i < HSLOTS; i += 64) {
    uint32_t s = i + lane;
    bool occ = s < HSLOTS && hk[s] != 0;
    uint64_t bal = __builtin_amdgcn_ballot_w64(occ);
    if (occ) skey[nd + mbcnt(bal)] = (uint64_t)hs[s] << 32 | s;
    nd += (uint32_t)__builtin_popcountll(bal);
  }
  uint32_t n2 = 1;
  while (n2 < nd) n2 <<= 1;
  for (uint32_t i = nd + lane; i < n2; i += 64) skey[i] = ~0ull;
  __builtin_amdgcn_wave_barrier();
  for (uint32_t k = 2; k <= n2; k <<= 1)
    for (uint32_t j = k >> 1; j > 0; j >>= 1) {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      __builtin_amdgcn_wave_barrier();
      for (uint32_t i = lane; i < n2; i += 64) {
        uint32_t ixj = i ^ j;
        if (ixj > i) {
          uint64_t x = skey[i], y = skey[ixj];
          bool up = (i & k) == 0;
          if ((x > y) == up) {
            skey[i] = y;
            skey[ixj] = x;
          }
        }
      }
    }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
  for (uint32_t i = lane; i < nd; i += 64) {
    uint32_t s = (uint32_t)skey[i];
    uint32_t key = hk[s];
    hh[i] = fnv_offset((int32_t)(key >> 16) - 32768, (int32_t)(key & 0xffffu) - 32768);
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
  // pos_of[k] = insertion index of the k-th iterated offset
  umap_order_par<64, HSLOTS>(hh, nd, okeys, seq_a, seq_b, bmin, WaveBarrier{}, pos_of, lane);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
  // pos_of[k] = insertion index at iteration position k; pick the best three by
  // (count desc, iteration position asc)
  uint32_t sel_pos[3];
  for (int rdx = 0; rdx < 3; ++rdx) {
    uint64_t m = 0;
    for (uint32_t kpos = lane; kpos < nd; kpos += 64) {
      bool used = false;
      for (int u = 0; u < rdx; ++u) used |= sel_pos[u] == kpos;
      if (!used) {
        uint32_t s = (uint32_t)skey[pos_of[kpos]];
        uint64_t cand = (uint64_t)hc[s] << 32 | (uint64_t)(0xffffffffu - kpos);
        m = cand > m ? cand : m;
      }
    }
    uint64_t w = wave_max_u64(m);
    sel_pos[rdx] = w ? 0xffffffffu - (uint32_t)w : 0xffffffffu;
    if (w != 0 && lane == 0) {
      uint32_t s = (uint32_t)skey[pos_of[sel_pos[rdx]]];
      ticket[2 * rdx] = hk[s];
      ticket[2 * rdx + 1] = hc[s];
    }
  }
}

// One wave per (pair, region).  Pair index q: current frame = q (q = 0 pairs the halo with
// frame 0), previous frame = q - 1 (pool slot n for the halo).
// Three instances behind kpm_match_pair, each walking the work list the one before it
// appended to (fixed grids, no host round trip).  Offset slots / shared codes / distinct
// offsets: TIER 0 = 256 / 256 / 128 (8 KB of LDS; the dense stream: ~110 shared codes but
// only a handful of distinct offsets per region), TIER 1 = 1024 / 1024 / 512 (39 KB),
// TIER 2 = 4096 / 2048 / 2048 (136 KB: one wave per CU), which flags the status word
// instead of passing on.
template <int TIER, uint32_t KM_HSLOTS, uint32_t KM_HSHIFT, uint32_t KM_MG, uint32_t KM_DMAX, uint32_t KM_BK>
__global__ __launch_bounds__(64) void kpm_match(KpmMatchParams p) {
  __shared__ uint32_t hk[KM_HSLOTS], hc[KM_HSLOTS], hs[KM_HSLOTS];  // key, count, first pair index
  // the shared-code list (steps a, b) and the tie path's sort scratch (after b) share storage:
  // LDS per wave decides how many (pair, region) items a CU has in flight
  constexpr uint32_t MG_BYTES = (3 * KM_MG + 1) * 4, TIE_BYTES = (8 + 8 + 4) * KM_DMAX;
  __shared__ __attribute__((aligned(8))) uint8_t shared_ab[MG_BYTES > TIE_BYTES ? MG_BYTES : TIE_BYTES];
  uint32_t* const mg_p = reinterpret_cast<uint32_t*>(shared_ab);
  uint32_t* const mg_c = mg_p + KM_MG;
  uint32_t* const mg_base = mg_c + KM_MG;
  uint64_t* const skey = reinterpret_cast<uint64_t*>(shared_ab);
  uint64_t* const hh = skey + KM_DMAX;
  uint32_t* const okeys = reinterpret_cast<uint32_t*>(hh + KM_DMAX);
  __shared__ uint16_t seq_a[KM_DMAX], seq_b[KM_DMAX], pos_of[KM_DMAX];
  __shared__ uint32_t bmin[KM_BK];

  const uint32_t lane = threadIdx.x;
  // TIER 0: one (pair, region) per workgroup.  Above: a fixed grid walks the work list the
  // tier below appended to (no launch of n*8 mostly empty workgroups).
  const uint32_t it_end = p.work_in[0];
  if (blockIdx.x == 0 && lane == 0 && it_end) kpm_mark_used(p.status, KPM_TIER_MATCH0 << TIER);
  for (uint32_t it = blockIdx.x; it < it_end; it += gridDim.x) {
  const uint32_t item = p.work_in[1 + it];
  const uint32_t q = item / p.nreg, r = item % p.nreg;
  uint32_t* ticket = p.tickets + ((size_t)q * p.nreg + r) * 6;
  __builtin_amdgcn_wave_barrier();
  if (lane < 6) ticket[lane] = 0;
  if (q == 0 && !p.has_halo) continue;
  const uint32_t stride = kpm_stride_u32(p.kmax);
  const uint32_t pslot = q == 0 ? p.n : q - 1;
  const uint32_t* P = p.pool + (size_t)pslot * stride;
  const uint32_t* Cc = p.pool + (size_t)q * stride;
  const uint32_t tstride = p.tstride ? p.tstride : p.kmax;
  const rmx_desc* pdesc = q == 0 ? p.halo_desc : p.desc + (size_t)(q - 1) * tstride;
  const rmx_desc* cdesc = p.desc + (size_t)q * tstride;
  const uint32_t* ph = P + r * 8;
  const uint32_t* ch = Cc + r * 8;
  const uint32_t Gc = ch[H_G];
  if (Gc == 0 || ph[H_G] == 0) continue;
  // kpm.hpp:219-222: all codes unless both regions are rich in weight-2 keypoints
  const bool use_all = ph[H_W2] < p.weight_switch || ch[H_W2] <= p.weight_switch;
  const uint32_t* c_gstart = Cc + kpm_off_gstart(p.kmax) + ch[H_GOFF] + r;
  const uint32_t* c_gfirst = Cc + kpm_off_gfirst(p.kmax) + ch[H_GOFF];
  const uint32_t* c_giter = Cc + kpm_off_giter(p.kmax) + ch[H_GOFF];
  const uint32_t* c_pts = Cc + kpm_off_pts(p.kmax);
  const uint32_t* p_gstart = P + kpm_off_gstart(p.kmax) + ph[H_GOFF] + r;
  const uint32_t* p_gfirst = P + kpm_off_gfirst(p.kmax) + ph[H_GOFF];
  const uint32_t* p_ht = P + kpm_off_ht(p.kmax) + ph[H_HTOFF];
  const uint32_t* p_pts = P + kpm_off_pts(p.kmax);
  const uint32_t pmask = ph[H_HTMASK];

  // ---- a. shared codes in the current map's iteration order, with pair-index bases ----
  // Every lookup is a chain of dependent loads (iteration order -> code -> hash slot -> code
  // of the other map -> group bounds).  Four groups per lane walk it side by side: the wave
  // is latency-bound (one group per lane per trip: 0.91 ms for the dense stream's 80 000 items).
  uint32_t nm = 0, total = 0;  // uniform
  bool overflow = false;
  const uint32_t c_goff = ch[H_GOFF], p_goff = ph[H_GOFF];
  for (uint32_t base = 0; base < Gc && !overflow; base += 256) {
    uint32_t gc[4], c[4][4], sl[4], gp[4], pairs[4], cstart[4], pstart[4], ncur[4];
    bool pend[4], hit[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t idx = base + 64 * u + lane;
      pend[u] = idx < Gc;
      gc[u] = pend[u] ? c_giter[idx] : 0u;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      c[u][0] = c[u][1] = c[u][2] = c[u][3] = 0;
      if (pend[u]) load_gcode(Cc, p.kmax, c_goff + gc[u], c[u]);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      pend[u] = pend[u] && (use_all || (c[u][3] & 0xfu) == 2u);
      sl[u] = gcode_hash(c[u]) & pmask;
      hit[u] = false;
      gp[u] = 0;
    }
    for (;;) {  // probe rounds: the pending lookups of all lanes advance together
      uint32_t v[4], c2[4][4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = pend[u] ? p_ht[sl[u]] : 0u;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        pend[u] = pend[u] && v[u] != 0;
        c2[u][0] = c2[u][1] = c2[u][2] = c2[u][3] = 0;
        if (pend[u]) load_gcode(P, p.kmax, p_goff + v[u] - 1, c2[u]);
      }
      bool more = false;
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (pend[u]) {
          if (same_code(c[u], c2[u])) {
            gp[u] = v[u] - 1;
            hit[u] = true;
            pend[u] = false;
          } else {
            sl[u] = (sl[u] + 1) & pmask;
            more = true;
          }
        }
      if (__builtin_amdgcn_ballot_w64(more) == 0) break;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      pairs[u] = 0;
      cstart[u] = pstart[u] = ncur[u] = 0;
      if (hit[u]) {
        const uint32_t p0 = p_gstart[gp[u]], p1 = p_gstart[gp[u] + 1], c0 = c_gstart[gc[u]], c1 = c_gstart[gc[u] + 1];
        pstart[u] = p0;
        cstart[u] = c0;
        ncur[u] = c1 - c0;
        pairs[u] = (p1 - p0) * ncur[u];
      }
    }
    // the shared-code list in iteration order: chunk by chunk of 64 groups
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (base + 64 * u >= Gc || overflow) continue;  // uniform
      uint64_t bal = __builtin_amdgcn_ballot_w64(pairs[u] != 0);
      uint32_t incl = pairs[u];
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        uint32_t t = (uint32_t)__shfl_up((int)incl, o);
        if (lane >= (uint32_t)o) incl += t;
      }
      uint32_t cnt = (uint32_t)__builtin_popcountll(bal);
      if (nm + cnt > KM_MG) {
        overflow = true;
        continue;
      }
      if (pairs[u] != 0) {
        uint32_t at = nm + mbcnt(bal);
        mg_p[at] = pstart[u];                   // first point of the previous map's group
        mg_c[at] = cstart[u] | ncur[u] << 18;  // first point (< 2^18) and size (<= 4096) of the current map's group
        mg_base[at] = total + incl - pairs[u];
      }
      total += (uint32_t)__shfl((int)incl, 63);
      nm += cnt;
    }
  }
  if (overflow) {
    if (lane == 0) {
      if (TIER < 2) {
        p.work_out[1 + atomicAdd(&p.work_out[0], 1u)] = item;
      } else if (p.big) {
        p.big[1 + atomicAdd(&p.big[0], 1u)] = item;
        atomicOr(&p.status[1], 128u);
      } else {
        atomicOr(&p.status[1], 2u);
      }
    }
    continue;
  }
  if (nm == 0) continue;
  if (lane == 0) mg_base[nm] = total;
  for (uint32_t s = lane; s < KM_HSLOTS; s += 64) {
    hk[s] = 0;
    hc[s] = 0;
    hs[s] = 0xffffffffu;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();

  // ---- b. count_offsets: every (previous point, current point) pair of every shared code ----
  // (four pairs per lane: their point loads are in flight together)
  uint32_t my_new = 0;
  for (uint32_t t0 = 0; t0 < total; t0 += 256) {
    uint32_t pi[4], ci[4];
    bool act[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const uint32_t t = t0 + 64 * u + lane;
      act[u] = t < total;
      pi[u] = ci[u] = 0;
      if (act[u]) {
        uint32_t lo = 0, hi = nm;  // largest g with mg_base[g] <= t
        while (hi - lo > 1) {
          uint32_t mid = (lo + hi) >> 1;
          if (mg_base[mid] <= t)
            lo = mid;
          else
            hi = mid;
        }
        const uint32_t cw = mg_c[lo], nc = cw >> 18, rel = t - mg_base[lo];
        const uint32_t ip = rel / nc, ic = rel - ip * nc;  // previous outer, current inner (kpm.hpp:94-95)
        pi[u] = mg_p[lo] + ip;
        ci[u] = (cw & 0x3ffffu) + ic;
      }
    }
    uint32_t pp[4], cp[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      pp[u] = act[u] ? p_pts[pi[u]] : 0u;
      cp[u] = act[u] ? c_pts[ci[u]] : 0u;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (!act[u]) continue;
      const uint32_t t = t0 + 64 * u + lane;
      int32_t dx = (int32_t)(pp[u] & 0xffffu) - (int32_t)(cp[u] & 0xffffu);
      int32_t dy = (int32_t)(pp[u] >> 16) - (int32_t)(cp[u] >> 16);
      uint32_t key = off_key(dx, dy);
      uint32_t s = (key * 2654435761u) >> KM_HSHIFT;
      for (uint32_t probes = 0;; ++probes) {
        uint32_t v = atomicCAS(&hk[s], 0u, key);
        if (v == 0u) ++my_new;
        if (v == 0u || v == key) {
          atomicAdd(&hc[s], 1u);
          atomicMin(&hs[s], t);
          break;
        }
        s = (s + 1) & (KM_HSLOTS - 1);
        if (probes > KM_HSLOTS) break;  // table full: flagged below through the distinct count
      }
    }
  }
  uint32_t D = my_new;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) D += (uint32_t)__shfl_xor((int)D, o);
  if (D > KM_DMAX) {
    if (lane == 0) {
      if (TIER < 2) {
        p.work_out[1 + atomicAdd(&p.work_out[0], 1u)] = item;
      } else if (p.big) {
        p.big[1 + atomicAdd(&p.big[0], 1u)] = item;
        atomicOr(&p.status[1], 128u);
      } else {
        atomicOr(&p.status[1], 4u);
      }
    }
    continue;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();

  // ---- c. top_offsets(total, 3): by count, ties by iteration order of the totalizator ----
  // top-4 counts to detect whether any tie can influence the ticket
  uint64_t best[4];
  {
    uint64_t taken_mask_lo = 0;  // per-lane: which of my slots were taken (slot index = lane + 64*i)
    for (int rdx = 0; rdx < 4; ++rdx) {
      uint64_t m = 0;
      for (uint32_t i = 0; i < KM_HSLOTS / 64; ++i) {
        uint32_t s = lane + 64 * i;
        if (hk[s] != 0 && !((taken_mask_lo >> i) & 1ull)) {
          uint64_t cand = (uint64_t)hc[s] << 32 | (uint64_t)(0xffffffffu - s);
          m = cand > m ? cand : m;
        }
      }
      uint64_t w = wave_max_u64(m);
      best[rdx] = w;
      if (w != 0) {
        uint32_t s = 0xffffffffu - (uint32_t)w;
        if ((s & 63u) == lane) taken_mask_lo |= 1ull << (s >> 6);
      }
    }
  }
  const uint32_t c1 = (uint32_t)(best[0] >> 32), c2 = (uint32_t)(best[1] >> 32), c3 = (uint32_t)(best[2] >> 32),
                 c4 = (uint32_t)(best[3] >> 32);
  const bool tie = (D >= 2 && c1 == c2) || (D >= 3 && c2 == c3) || (D >= 4 && c3 == c4);
#ifdef RMX_STATS
  if (lane == 0) {
    atomicAdd(&p.status[16 + TIER * 8], 1u);
    if (tie) atomicAdd(&p.status[17 + TIER * 8], 1u);
    atomicAdd(&p.status[18 + TIER * 8], D);
    atomicAdd(&p.status[19 + TIER * 8], total);
    atomicAdd(&p.status[20 + TIER * 8], nm);
    if (tie && D > 13) atomicAdd(&p.status[21 + TIER * 8], 1u);
    if (tie && D > 29) atomicAdd(&p.status[22 + TIER * 8], 1u);
    if (tie && D > 59) atomicAdd(&p.status[23 + TIER * 8], 1u);
  }
#endif
  if (!tie) {
    if (lane < 3 && lane < D) {
      uint32_t s = 0xffffffffu - (uint32_t)best[lane];
      ticket[2 * lane] = hk[s];
      ticket[2 * lane + 1] = hc[s];
    }
    continue;
  }
  exact_top3<KM_HSLOTS>(hk, hc, hs, skey, hh, okeys, seq_a, seq_b, pos_of, bmin, ticket, lane);
  }  // work items
}

// ---- pair-wave instance: ONE wave handles all eight regions of a (previous, current) pair --------------
// On the sparse stream a region holds ~10 codes and ~10 point pairs, so a wave per (pair,
// region) idles 50+ lanes and the 8x larger grid is pure launch/latency cost.  Here the lanes
// run over the concatenated code lists of the eight regions; every region has its own
// 32-slot offset table.  A pair with more than PW_MG codes, or a region with more than
// PW_DMAX distinct offsets, is passed on to the per-(pair, region) instances via the work list.
constexpr uint32_t PW_MG = 128, PW_SLOTS = 32, PW_DMAX = 16;  // 6 KB of LDS: 26 waves per CU (12 KB: 13)
constexpr uint32_t PW_SPL = PW_SLOTS / 8;                      // slots per lane in the selection step

__device__ __forceinline__ void push_items(uint32_t* work, bool mine, uint32_t item, uint32_t lane) {
  const uint64_t bal = __builtin_amdgcn_ballot_w64(mine);
  if (bal == 0) return;
  uint32_t base = 0;
  if (lane == 0) base = atomicAdd(&work[0], (uint32_t)__builtin_popcountll(bal));
  base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
  if (mine) work[1 + base + mbcnt(bal)] = item;
}

__global__ __launch_bounds__(64) void kpm_match_pair(KpmMatchParams p) {
  __shared__ uint32_t hk[8 * PW_SLOTS], hc[8 * PW_SLOTS], hs[8 * PW_SLOTS];
  __shared__ uint32_t mg_p[PW_MG], mg_c[PW_MG], mg_n[PW_MG], mg_base[PW_MG + 1];
  __shared__ uint64_t skey[PW_SLOTS], hh[PW_SLOTS];
  __shared__ uint16_t seq_a[PW_SLOTS], seq_b[PW_SLOTS], pos_of[PW_SLOTS];
  __shared__ uint32_t okeys[PW_SLOTS], bmin[32];  // (<= 16 distinct offsets: 13 or 29 buckets)

  const uint32_t lane = threadIdx.x, q = blockIdx.x;
  uint32_t* tick = p.tickets + (size_t)q * 48;
  if (lane < 48) tick[lane] = 0;
  if (q == 0 && !p.has_halo) return;
  const uint32_t stride = kpm_stride_u32(p.kmax);
  const uint32_t* P = p.pool + (size_t)(q == 0 ? p.n : q - 1) * stride;
  const uint32_t* Cc = p.pool + (size_t)q * stride;
  const uint32_t hp = P[lane], hcu = Cc[lane];  // header word: region lane >> 3, field lane & 7
  // code-list prefix over the regions that can produce a ticket (both maps non-empty)
  uint32_t gpre[9];
  gpre[0] = 0;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const uint32_t gc = (uint32_t)__builtin_amdgcn_readlane((int)hcu, r * 8 + H_G);
    const uint32_t gp = (uint32_t)__builtin_amdgcn_readlane((int)hp, r * 8 + H_G);
    gpre[r + 1] = gpre[r] + ((gc && gp) ? gc : 0u);
  }
  const uint32_t Gtot = gpre[8];
  if (Gtot == 0) return;
  {
    const uint32_t my_gc = (uint32_t)__shfl((int)hcu, (int)((lane & 7u) * 8 + H_G));
    const uint32_t my_gp = (uint32_t)__shfl((int)hp, (int)((lane & 7u) * 8 + H_G));
    if (Gtot > PW_MG) {
      push_items(p.work_out, lane < 8 && my_gc && my_gp, q * 8 + lane, lane);
      return;
    }
  }
  const uint32_t* c_giter = Cc + kpm_off_giter(p.kmax);
  const uint32_t* c_gstart = Cc + kpm_off_gstart(p.kmax);
  const uint32_t* p_gstart = P + kpm_off_gstart(p.kmax);
  const uint32_t* p_ht = P + kpm_off_ht(p.kmax);
  const uint32_t* c_pts = Cc + kpm_off_pts(p.kmax);
  const uint32_t* p_pts = P + kpm_off_pts(p.kmax);

  // ---- a. shared codes, region by region in each current map's iteration order ----
  // (Gtot <= 128: both chunks of 64 codes walk their chains of dependent loads side by side)
  uint32_t nm = 0, total = 0;
  {
    uint32_t rg[2], gcv[2], c[2][4], sl[2], pv[2], goff_c[2], goff_p[2], htoff[2], pmask[2];
    uint32_t pairs[2], ps[2], cs[2], nc[2];
    bool pend[2], hit[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const uint32_t k = min(64u * u + lane, Gtot - 1);
      uint32_t r = 0, jbase = 0;
#pragma unroll
      for (int q2 = 1; q2 < 8; ++q2)
        if (k >= gpre[q2]) {
          r = q2;
          jbase = gpre[q2];
        }
      // (a region without codes has zero width, so the last boundary <= k names k's region)
      rg[u] = r;
      goff_c[u] = (uint32_t)__shfl((int)hcu, (int)(r * 8 + H_GOFF));
      const uint32_t w2c = (uint32_t)__shfl((int)hcu, (int)(r * 8 + H_W2));
      goff_p[u] = (uint32_t)__shfl((int)hp, (int)(r * 8 + H_GOFF));
      htoff[u] = (uint32_t)__shfl((int)hp, (int)(r * 8 + H_HTOFF));
      pmask[u] = (uint32_t)__shfl((int)hp, (int)(r * 8 + H_HTMASK));
      const uint32_t w2p = (uint32_t)__shfl((int)hp, (int)(r * 8 + H_W2));
      pend[u] = 64u * u + lane < Gtot;
      // kpm.hpp:219-222: all codes unless both regions are rich in weight-2 keypoints
      hit[u] = w2p < p.weight_switch || w2c <= p.weight_switch;  // (use_all, until the codes are here)
      gcv[u] = pend[u] ? c_giter[goff_c[u] + (k - jbase)] : 0u;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      c[u][0] = c[u][1] = c[u][2] = c[u][3] = 0;
      if (pend[u]) load_gcode(Cc, p.kmax, goff_c[u] + gcv[u], c[u]);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      pend[u] = pend[u] && (hit[u] || (c[u][3] & 0xfu) == 2u);
      hit[u] = false;
      sl[u] = gcode_hash(c[u]) & pmask[u];
      pv[u] = 0;
    }
    for (;;) {  // probe rounds: the pending lookups of all lanes advance together
      uint32_t v[2], c2[2][4];
#pragma unroll
      for (int u = 0; u < 2; ++u) v[u] = pend[u] ? p_ht[htoff[u] + sl[u]] : 0u;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        pend[u] = pend[u] && v[u] != 0;
        c2[u][0] = c2[u][1] = c2[u][2] = c2[u][3] = 0;
        if (pend[u]) load_gcode(P, p.kmax, goff_p[u] + v[u] - 1, c2[u]);
      }
      bool more = false;
#pragma unroll
      for (int u = 0; u < 2; ++u)
        if (pend[u]) {
          if (same_code(c[u], c2[u])) {
            pv[u] = v[u];
            hit[u] = true;
            pend[u] = false;
          } else {
            sl[u] = (sl[u] + 1) & pmask[u];
            more = true;
          }
        }
      if (__builtin_amdgcn_ballot_w64(more) == 0) break;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      pairs[u] = ps[u] = cs[u] = nc[u] = 0;
      if (hit[u]) {
        const uint32_t p0 = p_gstart[goff_p[u] + rg[u] + pv[u] - 1], p1 = p_gstart[goff_p[u] + rg[u] + pv[u]];
        const uint32_t c0 = c_gstart[goff_c[u] + rg[u] + gcv[u]], c1 = c_gstart[goff_c[u] + rg[u] + gcv[u] + 1];
        ps[u] = p0;
        cs[u] = c0;
        nc[u] = c1 - c0;
        pairs[u] = (p1 - p0) * nc[u];
      }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (64u * u >= Gtot) continue;  // uniform
      const uint64_t bal = __builtin_amdgcn_ballot_w64(pairs[u] != 0);
      uint32_t incl = pairs[u];
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const uint32_t t = (uint32_t)__shfl_up((int)incl, o);
        if (lane >= (uint32_t)o) incl += t;
      }
      if (pairs[u] != 0) {
        const uint32_t at = nm + mbcnt(bal);
        mg_p[at] = ps[u];
        mg_c[at] = cs[u] | rg[u] << 20;
        mg_n[at] = nc[u];
        mg_base[at] = total + incl - pairs[u];
      }
      total += (uint32_t)__shfl((int)incl, 63);
      nm += (uint32_t)__builtin_popcountll(bal);
    }
  }
  if (nm == 0) return;
  if (lane == 0) mg_base[nm] = total;
  for (uint32_t s = lane; s < 8 * PW_SLOTS; s += 64) {
    hk[s] = 0;
    hc[s] = 0;
    hs[s] = 0xffffffffu;
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();

  // ---- b. count_offsets: previous outer, current inner (kpm.hpp:94-95), one table per region ----
  uint32_t fail = 0;
  for (uint32_t t0 = 0; t0 < total; t0 += 128) {  // two pairs per lane: their point loads in flight together
    uint32_t pi[2], ci[2], rr2[2];
    bool act[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const uint32_t t = t0 + 64u * u + lane;
      act[u] = t < total;
      pi[u] = ci[u] = rr2[u] = 0;
      if (act[u]) {
        uint32_t lo = 0, hi = nm;
        while (hi - lo > 1) {
          const uint32_t mid = (lo + hi) >> 1;
          if (mg_base[mid] <= t)
            lo = mid;
          else
            hi = mid;
        }
        const uint32_t nc = mg_n[lo], rel = t - mg_base[lo], cw = mg_c[lo];
        const uint32_t ip = rel / nc, ic = rel - ip * nc;
        rr2[u] = cw >> 20;
        pi[u] = mg_p[lo] + ip;
        ci[u] = (cw & 0xfffffu) + ic;
      }
    }
    uint32_t pp[2], cp[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      pp[u] = act[u] ? p_pts[pi[u]] : 0u;
      cp[u] = act[u] ? c_pts[ci[u]] : 0u;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if (!act[u]) continue;
      const uint32_t t = t0 + 64u * u + lane, r = rr2[u];
      const int32_t dx = (int32_t)(pp[u] & 0xffffu) - (int32_t)(cp[u] & 0xffffu);
      const int32_t dy = (int32_t)(pp[u] >> 16) - (int32_t)(cp[u] >> 16);
      const uint32_t key = off_key(dx, dy), tb = r * PW_SLOTS;
      uint32_t s = (key * 2654435761u) >> 27;
      for (uint32_t probes = 0;; ++probes) {
        const uint32_t v = atomicCAS(&hk[tb + s], 0u, key);
        if (v == 0u || v == key) {
          atomicAdd(&hc[tb + s], 1u);
          atomicMin(&hs[tb + s], t);
          break;
        }
        s = (s + 1) & (PW_SLOTS - 1);
        if (probes >= PW_SLOTS) {
          fail |= 1u << r;
          break;
        }
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) fail |= (uint32_t)__shfl_xor((int)fail, o);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
  const uint32_t rr = lane >> 3, sub = lane & 7u;
  uint32_t D = 0;  // distinct offsets of region rr
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const uint32_t d =
        (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(lane < PW_SLOTS && hk[r * PW_SLOTS + lane] != 0));
    if (d > PW_DMAX) fail |= 1u << r;
    if (rr == (uint32_t)r) D = d;
  }
  push_items(p.work_out, lane < 8 && ((fail >> lane) & 1u), q * 8 + lane, lane);

  // ---- c. top_offsets(total, 3), eight lanes per region ----
  const bool okr = !((fail >> rr) & 1u) && D > 0;
  const uint32_t tb = rr * PW_SLOTS;
  uint64_t best[4];
  uint32_t taken = 0;
#pragma unroll
  for (int rd = 0; rd < 4; ++rd) {
    uint64_t m = 0;
#pragma unroll
    for (uint32_t i = 0; i < PW_SPL; ++i) {
      const uint32_t s = sub * PW_SPL + i;
      if (hk[tb + s] != 0 && !((taken >> i) & 1u)) {
        const uint64_t cand = (uint64_t)hc[tb + s] << 32 | (uint64_t)(0xffffffffu - s);
        m = cand > m ? cand : m;
      }
    }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {
      const uint64_t t = (uint64_t)__shfl_xor((long long)m, o);
      m = t > m ? t : m;
    }
    best[rd] = m;
    if (m != 0) {
      const uint32_t s = 0xffffffffu - (uint32_t)m;
      if (s / PW_SPL == sub) taken |= 1u << (s % PW_SPL);
    }
  }
  const uint32_t c1 = (uint32_t)(best[0] >> 32), c2 = (uint32_t)(best[1] >> 32), c3 = (uint32_t)(best[2] >> 32),
                 c4 = (uint32_t)(best[3] >> 32);
  const bool tie = (D >= 2 && c1 == c2) || (D >= 3 && c2 == c3) || (D >= 4 && c3 == c4);
  if (okr && !tie && sub < 3 && sub < D) {
    const uint32_t s = 0xffffffffu - (uint32_t)best[sub];
    tick[rr * 6 + 2 * sub] = hk[tb + s];
    tick[rr * 6 + 2 * sub + 1] = hc[tb + s];
  }
  uint64_t todo = __builtin_amdgcn_ballot_w64(okr && tie && sub == 0);
  while (todo) {  // regions whose ticket depends on the totalizator's iteration order (uniform loop)
    const uint32_t r = (uint32_t)__builtin_ctzll(todo) >> 3;
    todo &= todo - 1;
    exact_top3<PW_SLOTS>(hk + r * PW_SLOTS, hc + r * PW_SLOTS, hs + r * PW_SLOTS, skey, hh, okeys, seq_a, seq_b, pos_of,
                         bmin, tick + r * 6, lane);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
  }
}

struct KpmDeclareParams {
  const uint32_t* pool;
  const uint32_t* tickets;
  int32_t* dxdy;
  uint8_t* matched;
  uint32_t n, kmax, nreg, has_halo, region_votes;
  // instances of the chain that were NOT launched in this step (KPM_TIER_* bits; 0 = all were): a frame whose
  // tier word still names one of them, or a work list left for one, turns the step into one rmx_kpm_resolve
  // has to redo (status[1] |= 128, status[14] = which)
  uint32_t skipped;
  const uint32_t* tier;  // n + has_halo tier words (kpm_build_small)
  const uint32_t* work;  // the three work lists' counters, work_stride words apart
  uint32_t work_stride;
  uint32_t* status;
};

// kpm.hpp:395-415 tail: get_active, count (Borda), top_offsets(…, 2), declare.
__global__ void kpm_declare(KpmDeclareParams p) {
  const uint32_t q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= p.n) return;
  if (p.skipped) {
    auto waits_for = [](uint32_t tw) -> uint32_t {
      return tw == 4u ? KPM_TIER_SMALL64 : tw == 1u ? KPM_TIER_MEDIUM256 : tw == 2u ? KPM_TIER_MEDIUM512 : (tw == 3u || tw == 5u) ? KPM_TIER_GENERAL : 0u;
    };
    uint32_t miss = waits_for(p.tier[q]);
    if (q == 0) {
      if (p.has_halo) miss |= waits_for(p.tier[p.n]);
      for (uint32_t t = 0; t < 3; ++t)
        if (p.work[t * p.work_stride]) miss |= KPM_TIER_MATCH0 << t;
    }
    miss &= p.skipped;
    if (miss) {
      atomicOr(&p.status[14], miss);
      atomicOr(&p.status[1], 128u);
    }
  }
  int32_t dx = 0, dy = 0;
  uint8_t ok = 0;
  if (q > 0 || p.has_halo) {
    const uint32_t* Cc = p.pool + (size_t)q * kpm_stride_u32(p.kmax);
    uint32_t active = 0;
    for (uint32_t r = 0; r < p.nreg; ++r) active += Cc[r * 8 + H_E] != 0;
    if (active >= p.nreg / 4) {
      uint32_t keys[24], tot[24], nk = 0;
      const uint32_t* t = p.tickets + (size_t)q * p.nreg * 6;
      for (uint32_t r = 0; r < p.nreg; ++r) {
        uint32_t rank = p.region_votes;
        for (uint32_t j = 0; j < 3; ++j, --rank) {
          uint32_t key = t[r * 6 + 2 * j];
          if (key == 0) break;
          uint32_t i = 0;
          for (; i < nk; ++i)
            if (keys[i] == key) break;
          if (i == nk) {
            keys[nk] = key;
            tot[nk] = 0;
            ++nk;
          }
          tot[i] += rank;
        }
      }
      if (nk > 0) {
        uint32_t b0 = 0;
        for (uint32_t i = 1; i < nk; ++i)
          if (tot[i] > tot[b0]) b0 = i;
        uint32_t second = 0;
        bool has2 = false;
        for (uint32_t i = 0; i < nk; ++i)
          if (i != b0) {
            has2 = true;
            second = tot[i] > second ? tot[i] : second;
          }
        if (!(has2 && tot[b0] < second + active / 2)) {
          ok = 1;
          dx = (int32_t)(keys[b0] >> 16) - 32768;
          dy = (int32_t)(keys[b0] & 0xffffu) - 32768;
        }
      }
    }
  }
  p.dxdy[2 * q] = dx;
  p.dxdy[2 * q + 1] = dy;
  p.matched[q] = ok;
}

}  // namespace rmx
