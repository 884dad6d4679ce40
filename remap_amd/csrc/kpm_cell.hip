// kpm_cell.hip — cellular variant of kpm::match (reference kpm.hpp:371-393) used by
// fgs::match_partial / match_all (fgs.hpp:119-140) to pair fragments.
//
// For a pair (previous fragment, current fragment), both described by single-region
// descriptor tables (grid 1x1, overlap 0) already turned into region tables by kpm_build:
//   count_offsets (kpm.hpp:231-264): every point pair of every shared code votes for
//     offset = previous - current and marks the cell (min(px,cx)/cw, min(py,cy)/ch);
//   find_best (kpm.hpp:281-298): the offset with most votes — the FIRST such offset in the
//     iteration order of the std::unordered_map<offset, ...> totalizator on ties, which is
//     replayed with umap_order_par exactly as in kpm_match;
//   count_active_cells / filter_keypoints (kpm.hpp:302-367): distinct cells of the current
//     keypoints inside the overlap whose translated position is covered by the previous mask;
//   accept unless matched_cells < active * 0.66f (kpm.hpp:388) — the path's only fp32 op,
//     evaluated with the same two IEEE single operations.
// One wave per pair; all tables in LDS.  A pair that outgrows them (more than KC_DMAX distinct
// offsets, a full (offset, cell) set) is marked valid = 2 and matched by fgs_big.hip.
#include "common.h"

namespace rmx {

struct KpmCellParams {
  const rmx_desc* desc;          // nfrag tables, stride kmax
  const uint32_t* pool;          // region tables, one region per fragment
  const uint8_t* const* masks;   // device array: blend mask of every fragment
  const uint32_t* dims;          // nfrag x {width, height}
  const uint32_t* pairs;         // npairs x {previous, current}
  rmx_vote* votes;
  uint32_t* status;
  uint32_t nfrag, npairs, kmax, cell_w, cell_h;
};

constexpr uint32_t KC_HS = 2048;    // offset slots
constexpr uint32_t KC_PS = 8192;    // (offset, cell) slots
constexpr uint32_t KC_MG = 1024;    // shared codes
constexpr uint32_t KC_DMAX = 1024;  // distinct offsets

__global__ __launch_bounds__(64) void kpm_cell_match(KpmCellParams p) {
  __shared__ uint32_t hk[KC_HS], hc[KC_HS], hs[KC_HS], hcell[KC_HS];  // key, votes, first pair index, #cells
  __shared__ unsigned long long pc[KC_PS];                             // (offset slot + 1) << 32 | cell key
  __shared__ uint32_t mg_p[KC_MG], mg_c[KC_MG], mg_base[KC_MG + 1];
  __shared__ uint64_t hh[KC_DMAX], skey[KC_DMAX];
  __shared__ uint16_t seq_a[KC_DMAX], seq_b[KC_DMAX], pos_of[KC_DMAX];
  __shared__ uint32_t okeys[KC_DMAX], bmin[1109];

  const uint32_t lane = threadIdx.x;
  const uint32_t q = blockIdx.x;
  rmx_vote* out = p.votes + q;
  if (lane == 0) *out = rmx_vote{0, 0, 0, 0};
  const uint32_t fp = p.pairs[2 * q], fc = p.pairs[2 * q + 1];
  const uint32_t stride = kpm_stride_u32(p.kmax);
  const uint32_t* P = p.pool + (size_t)fp * stride;
  const uint32_t* Cc = p.pool + (size_t)fc * stride;
  const rmx_desc* pdesc = p.desc + (size_t)fp * p.kmax;
  const rmx_desc* cdesc = p.desc + (size_t)fc * p.kmax;
  const uint32_t Gc = Cc[H_G];
  if (Gc == 0 || P[H_G] == 0) return;
  const uint32_t* c_gstart = Cc + kpm_off_gstart(p.kmax) + Cc[H_GOFF];
  const uint32_t* c_gfirst = Cc + kpm_off_gfirst(p.kmax) + Cc[H_GOFF];
  const uint32_t* c_giter = Cc + kpm_off_giter(p.kmax) + Cc[H_GOFF];
  const uint32_t* c_pts = Cc + kpm_off_pts(p.kmax);
  const uint32_t* p_gstart = P + kpm_off_gstart(p.kmax) + P[H_GOFF];
  const uint32_t* p_gfirst = P + kpm_off_gfirst(p.kmax) + P[H_GOFF];
  const uint32_t* p_ht = P + kpm_off_ht(p.kmax) + P[H_HTOFF];
  const uint32_t* p_pts = P + kpm_off_pts(p.kmax);
  const uint32_t pmaskh = P[H_HTMASK];

  for (uint32_t s = lane; s < KC_HS; s += 64) {
    hk[s] = 0;
    hc[s] = 0;
    hs[s] = 0xffffffffu;
    hcell[s] = 0;
  }
  for (uint32_t s = lane; s < KC_PS; s += 64) pc[s] = 0ull;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();

  // The current map's codes are walked in its iteration order, KC_MG codes at a time: (a) the
  // shared ones of the chunk with their pair counts, (b) the votes of their point pairs.  The
  // pair index t (first touch of an offset = its insertion into the totalizator) runs on
  // across chunks, so a fragment may share any number of codes with the other one.
  uint32_t my_new = 0, my_fail = 0, pairs_before = 0;
  for (uint32_t cb = 0; cb < Gc; cb += KC_MG) {
    const uint32_t ce = min(Gc, cb + KC_MG);
    // ---- a. shared codes in the current map's iteration order (kpm.hpp:257-261) ----
    uint32_t nm = 0, total = 0;
    for (uint32_t base = cb; base < ce; base += 64) {
      uint32_t pairs = 0, gp = 0, gc = 0;
      if (base + lane < ce) {
        gc = c_giter[base + lane];
        uint32_t c[4];
        load_code(cdesc + c_gfirst[gc], c);
        uint32_t s = fnv_code_lo(c) & pmaskh;
        for (;;) {
          uint32_t v = p_ht[s];
          if (v == 0) break;
          uint32_t c2[4];
          load_code(pdesc + p_gfirst[v - 1], c2);
          if (same_code(c, c2)) {
            gp = v - 1;
            pairs = (p_gstart[gp + 1] - p_gstart[gp]) * (c_gstart[gc + 1] - c_gstart[gc]);
            break;
          }
          s = (s + 1) & pmaskh;
        }
      }
      uint64_t bal = __builtin_amdgcn_ballot_w64(pairs != 0);
      uint32_t incl = pairs;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        uint32_t t = (uint32_t)__shfl_up((int)incl, o);
        if (lane >= (uint32_t)o) incl += t;
      }
      if (pairs != 0) {
        uint32_t at = nm + mbcnt(bal);
        mg_p[at] = gp;
        mg_c[at] = gc;
        mg_base[at] = total + incl - pairs;
      }
      total += (uint32_t)__shfl((int)incl, 63);
      nm += (uint32_t)__builtin_popcountll(bal);
    }
    if (nm == 0) continue;
    if (lane == 0) mg_base[nm] = total;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();

    // ---- b. votes per offset and distinct cells per offset ----
    for (uint32_t t = lane; t < total; t += 64) {
      uint32_t lo = 0, hi = nm;
      while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (mg_base[mid] <= t)
          lo = mid;
        else
          hi = mid;
      }
      uint32_t gp = mg_p[lo], gc = mg_c[lo], rel = t - mg_base[lo];
      uint32_t cs = c_gstart[gc], nc = c_gstart[gc + 1] - cs;
      uint32_t ip = rel / nc, ic = rel - ip * nc;
      uint32_t pp = p_pts[p_gstart[gp] + ip], cp = c_pts[cs + ic];
      int32_t px = (int32_t)(pp & 0xffffu), py = (int32_t)(pp >> 16), cx = (int32_t)(cp & 0xffffu), cy = (int32_t)(cp >> 16);
      uint32_t key = off_key(px - cx, py - cy);
      uint32_t cellk = (uint32_t)(min(px, cx) / (int32_t)p.cell_w) << 12 | (uint32_t)(min(py, cy) / (int32_t)p.cell_h);
      uint32_t s = (key * 2654435761u) >> 21;
      bool placed = false;
      for (uint32_t probes = 0; probes <= KC_HS; ++probes) {
        uint32_t v = atomicCAS(&hk[s], 0u, key);
        if (v == 0u) ++my_new;
        if (v == 0u || v == key) {
          atomicAdd(&hc[s], 1u);
          atomicMin(&hs[s], pairs_before + t);
          placed = true;
          break;
        }
        s = (s + 1) & (KC_HS - 1);
      }
      if (!placed) {
        ++my_fail;
        continue;
      }
      unsigned long long k64 = (unsigned long long)(s + 1) << 32 | cellk;
      uint32_t h = (uint32_t)((k64 * 0x9E3779B97F4A7C15ull) >> 51);  // 13 bits
      bool done = false;
      for (uint32_t probes = 0; probes <= KC_PS; ++probes) {
        unsigned long long v = atomicCAS(&pc[h], 0ull, k64);
        if (v == 0ull) atomicAdd(&hcell[s], 1u);
        if (v == 0ull || v == k64) {
          done = true;
          break;
        }
        h = (h + 1) & (KC_PS - 1);
      }
      if (!done) ++my_fail;
    }
    pairs_before += total;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();  // the chunk's lists are free again
  }
  if (pairs_before == 0) return;  // offsets.empty() -> nullopt (kpm.hpp:380-382)
  uint32_t D = my_new, F = my_fail;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    D += (uint32_t)__shfl_xor((int)D, o);
    F += (uint32_t)__shfl_xor((int)F, o);
  }
  if (D > KC_DMAX || F != 0) {  // does not fit these tables: the host sends the pair to fgs_big.hip
    if (lane == 0) out->valid = 2u;
    return;
  }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();

  // ---- c. find_best: most votes, first in totalizator iteration order on ties ----
  uint64_t best2[2];
  {
    uint32_t taken = 0xffffffffu;
    for (int rdx = 0; rdx < 2; ++rdx) {
      uint64_t m = 0;
      for (uint32_t i = 0; i < KC_HS / 64; ++i) {
        uint32_t s = lane + 64 * i;
        if (hk[s] != 0 && s != taken) {
          uint64_t cand = (uint64_t)hc[s] << 32 | (uint64_t)(0xffffffffu - s);
          m = cand > m ? cand : m;
        }
      }
      best2[rdx] = wave_max_u64(m);
      taken = 0xffffffffu - (uint32_t)best2[rdx];
    }
  }
  uint32_t best_slot = 0xffffffffu - (uint32_t)best2[0];
  const uint32_t cmax = (uint32_t)(best2[0] >> 32);
  if (D >= 2 && (uint32_t)(best2[1] >> 32) == cmax) {
    // tie at the top: replay the totalizator's node order (insertion = first touch)
    uint32_t nd = 0;
    for (uint32_t i = 0; i < KC_HS; i += 64) {
      uint32_t s = i + lane;
      bool occ = hk[s] != 0;
      uint64_t bal = __builtin_amdgcn_ballot_w64(occ);
      if (occ) skey[nd + mbcnt(bal)] = (uint64_t)hs[s] << 32 | s;
      nd += (uint32_t)__builtin_popcountll(bal);
    }
    uint32_t n2 = 1;
    while (n2 < nd) n2 <<= 1;
    for (uint32_t i = nd + lane; i < n2; i += 64) skey[i] = ~0ull;
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
      uint32_t key = hk[(uint32_t)skey[i]];
      hh[i] = fnv_offset((int32_t)(key >> 16) - 32768, (int32_t)(key & 0xffffu) - 32768);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    umap_order_par<64>(hh, nd, okeys, seq_a, seq_b, bmin, WaveBarrier{}, pos_of, lane);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
    __builtin_amdgcn_wave_barrier();
    uint32_t first = 0xffffffffu;  // smallest iteration position holding cmax votes
    for (uint32_t kpos = lane; kpos < nd; kpos += 64)
      if (hc[(uint32_t)skey[pos_of[kpos]]] == cmax) first = min(first, kpos);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) first = min(first, (uint32_t)__shfl_xor((int)first, o));
    best_slot = (uint32_t)skey[pos_of[first]];
  }
  const uint32_t bkey = hk[best_slot], matched_cells = hcell[best_slot];
  const int32_t bdx = (int32_t)(bkey >> 16) - 32768, bdy = (int32_t)(bkey & 0xffffu) - 32768;

  // ---- d. active cells (kpm.hpp:302-367; only the current-side limits are used) ----
  const uint32_t pw = p.dims[2 * fp], ph = p.dims[2 * fp + 1], cw = p.dims[2 * fc], chh = p.dims[2 * fc + 1];
  auto limits = [](int32_t delta, uint64_t previous, uint64_t current, uint64_t& lo, uint64_t& hi) {
    if (delta < 0) {
      uint64_t d = (uint64_t)(-(int64_t)delta);
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
  __builtin_amdgcn_wave_barrier();
  for (uint32_t s = lane; s < KC_PS; s += 64) pc[s] = 0ull;
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
  __builtin_amdgcn_wave_barrier();
  const uint8_t* pm = p.masks[fp];
  const uint32_t ec = Cc[H_E];
  uint32_t act = 0, fail2 = 0;
  for (uint32_t i = lane; i < ec; i += 64) {
    uint32_t cp = c_pts[Cc[H_POFF] + i];
    uint64_t x = cp & 0xffffu, y = cp >> 16;
    if (x >= left && x < right && y >= top && y < bottom) {
      int64_t idx = (int64_t)pw * ((int64_t)y + bdy) + ((int64_t)x + bdx);
      if (pm[idx] != 0) {
        uint32_t ox = (uint32_t)((int32_t)(x - left) / (int32_t)p.cell_w), oy = (uint32_t)((int32_t)(y - top) / (int32_t)p.cell_h);
        unsigned long long k64 = 1ull << 32 | (unsigned long long)(ox << 12 | oy);
        uint32_t h = (uint32_t)((k64 * 0x9E3779B97F4A7C15ull) >> 51);
        bool done = false;
        for (uint32_t probes = 0; probes <= KC_PS; ++probes) {
          unsigned long long v = atomicCAS(&pc[h], 0ull, k64);
          if (v == 0ull) ++act;
          if (v == 0ull || v == k64) {
            done = true;
            break;
          }
          h = (h + 1) & (KC_PS - 1);
        }
        if (!done) ++fail2;
      }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    act += (uint32_t)__shfl_xor((int)act, o);
    fail2 += (uint32_t)__shfl_xor((int)fail2, o);
  }
  if (fail2) {
    if (lane == 0) out->valid = 2u;
    return;
  }
  // ---- e. kpm.hpp:388 ----
  if (lane == 0) {
    const float lhs = (float)matched_cells, rhs = (float)act * 0.66f;
    if (!(lhs < rhs)) *out = rmx_vote{bdx, bdy, cmax, 1u};
  }
}

}  // namespace rmx
