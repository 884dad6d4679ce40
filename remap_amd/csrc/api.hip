// api.hip — C ABI of remap_amd (include/remap_amd.h): context, workspace, kernel launches.
// One translation unit: the kernel files are included so a single hipcc invocation builds
// libremap_amd.so for gfx950.
#include <hip/hip_runtime.h>
#include <memory>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "common.h"
#include "kpe.hip"
#include "kpe_bs.hip"
#include "kpe_pc.hip"
#include "kpe_fused.hip"
#include "table.hip"
#include "kpm.hip"
#include "kpm_cell.hip"
#include "fgs_big.hip"
#include "fgm.hip"
#include "fgm_bs.hip"
#include "nic.hip"
#include "fdf.hip"

using namespace rmx;

struct rmx_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;  // created by rmx_ctx_create (RMX_STREAM_OWN), destroyed with the context
  uint32_t W = 0, H = 0, gw = 0, gh = 0, overlap = 0, kmax = 0;
  Bands bands{};
  // workspace (device)
  RawDesc* raw = nullptr;
  uint32_t* kraw = nullptr;    // per-frame append counters of the current batch: one of two buffers (2 x n_cap)
  uint32_t* kraw_base = nullptr;   // (the allocation)
  uint32_t* kraw_other = nullptr;  // the other one; its first kraw_clean entries are zero (cleared by the table kernels)
  uint32_t kraw_clean = 0;
  uint32_t* tflag = nullptr;  // per frame: left to kpe_table_rest
  uint32_t* cand = nullptr;  // candidate bit plane, n * ceil(H/8) * W/4 dwords
  uint32_t* tab_ws = nullptr;  // kpe_table_big (kmax > 16 384): 2 * ceil(W*H/32) dwords per frame
  uint32_t n_cap = 0;
  uint32_t n_last = 0;       // frames of the last rmx_kpe_batch
  size_t lds_table_rest = 64 * 1024, lds_fused = 64 * 1024;  // dynamic LDS granted to kpe_table_rest / kpe_fused on this context's device
  uint32_t rps_max = 48;       // longest strip (rows, multiple of 8)
  bool kpe_bs = false;         // 3x3 pass on bit planes (kpe_bs.hip): widths that are multiples of 16 (RMX_KPE_BS=0: kpe_mf)
  uint32_t bs_ra = 12;         // rows per strip of kpe_bs (two strips per workgroup)
  uint32_t bs_ra_ring = 24;    // ... of kpe_bs_ring (its finish does not care how far behind the stream it runs)
  // Which finish the bit-plane extraction kernel uses.  kpe_bs_ring (the finish rolled into the row loop, windows
  // from an LDS ring) is 8 % faster on streams with few keypoints and 20 % slower on dense ones (1 100 per
  // 320x240 frame): chosen when the last batch whose status words came back had at most one keypoint per 320
  // pixels (kpe_table samples every 32nd frame's count into status[15]); RMX_BS_RING=0/1 forces it.
  int bs_ring_forced = -1;
  // kpe_pc (kpe_pc.hip: candidate test by pairwise comparison, medians patched in the LDS ring) takes the ring
  // kernel's place where a frame's rows fill a wave: widths that are multiples of 32 whose 32-pixel chunks divide
  // 64 with at most eight lanes idle.  RMX_KPE_PC=0: kpe_bs_ring.  pc_ra: rows per strip (one strip per wave).
  bool kpe_pc = false;
  bool pc_dense = false;  // kpe_pc<8> (set with bs_ring by the density of the batch before; RMX_PC_DENSE=0/1 forces it)
  int pc_dense_forced = -1;
  uint32_t pc_ra = 30, pc_ra_short = 30, pc_long_pct = 0;  // RMX_PC_PLAN=long,short,percent
  uint32_t* pc_ovf = nullptr;  // PC_OVF_DWORDS per wave of the launch, all zero between launches
  size_t pc_ovf_bytes = 0;
  const char* kpe_kernel = "";  // the extraction kernel rmx_kpe_batch launched last (rmx_kpe_kernel)
  bool bs_ring = false;
  // rmx_kpm_seq's launch set (KPM_TIER_* bits, kpm.hip).  Adaptive: the first step of the context launches every
  // instance of the build / match chains and learns which found work; later steps launch those only — an instance
  // with 50-140 KB of LDS that has nothing to do still waits for a CU to drain when another stream's extraction
  // kernel fills the chip, and the whole chain behind it.  A step that needed one that was not launched reports
  // flag 128 and is redone by rmx_kpm_resolve with all of them; the set grows by what that pass used.
  // contexts of one pipeline (blocks of a stream in flight on streams of their own) whose extraction kernels
  // take turns: a context waits for the extraction kernel the group launched last and leaves its own as the
  // one to wait for — recorded right behind the extraction kernel, before the table kernels (rmx_kpe_take_turns)
  struct KpeTurns { hipEvent_t last = nullptr; };
  std::shared_ptr<KpeTurns> turns;
  hipEvent_t turn_ev = nullptr;
  bool kpm_adaptive = false;
  bool kpm_learned = false;
  uint32_t kpm_tiers = 127u;
  // launch shape of kpe_fused for this context's frames (plan_fused); ok = false: kpe_main + kpe_finish
  struct FusedPlan {
    bool ok = false, v16 = false;
    uint32_t R = 0, nt = 0, fpb = 0, pitch = 0, lds = 0;
  } fused;
  uint32_t overlap_subs = 0;    // RMX_OVERLAP: finish of sub-batch i on a second stream beside main of i + 1
  hipStream_t aux = nullptr;
  std::vector<hipEvent_t> ev_main, ev_fin;
  uint32_t sub_mib = 1u << 20;  // sub-batch footprint (MiB); default: whole batch in one launch (measured best)
  uint32_t* status = nullptr;  // [0] kpe overflow, [1] kpm capacity, [2..] scratch
  // status words as read back by the last rmx_frc_scan (which synchronises anyway); valid
  // until the next call that can raise them — saves two host round trips per step
  uint32_t* host_words = nullptr;  // pinned, 16 words
  uint32_t cached_status[2] = {0, 0};
  bool status_cached = false;
  KpmWorkspace kpm{};
  uint32_t* blit_ws = nullptr;  // fgm_blit binning workspace
  void* pack_buf = nullptr;     // staging of rmx_download_2d
  size_t pack_cap = 0;
  void* fdf_ws = nullptr;       // foreground-filter labelling workspace for fdf_cap frames
  uint32_t fdf_cap = 0;
  size_t blit_ws_bytes = 0;
  size_t ws_bytes = 0;
  // timing: HIP event pairs recorded on the context stream around each kernel launch,
  // resolved (averaged) only when rmx_last_kernel_ms is called — no host sync in between
  int timing = 0;  // 0 off, 1 every kernel, 2 kpe_main only (the roofline kernel)
  struct Track {
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;
    size_t used = 0;
  };
  Track ms_kpe, ms_finish, ms_table, ms_kpm_build, ms_kpm_match, ms_blit, ms_nic, ms_fdf;
  std::string err;
};

namespace {

constexpr uint32_t KPE_KMAX_LDS = 16384;      // tables the LDS instances of kpe_table sort
constexpr uint32_t KPE_KMAX_MAX = 1u << 22;   // rows per table a context may ask for (kpe_table_big above KPE_KMAX_LDS)

#define RMX_HIP(call)                                                                    \
  do {                                                                                   \
    hipError_t e_ = (call);                                                              \
    if (e_ != hipSuccess) {                                                              \
      ctx->err = std::string(#call) + ": " + hipGetErrorString(e_);                      \
      return RMX_ERR_DEVICE;                                                             \
    }                                                                                    \
  } while (0)

Bands make_bands(uint32_t W, uint32_t H, uint32_t gw, uint32_t gh, uint32_t ov) {
  Bands b{};
  b.gw = gw;
  b.gh = gh;
  uint32_t rw = W / gw - ov / 2, rh = H / gh - ov / 2;
  for (uint32_t c = 0; c < gw; ++c) {
    uint32_t s = 2 + c * (rw + ov);
    b.x0[c] = (uint16_t)(c ? s - ov : s);
    b.x1[c] = (uint16_t)((c + 1 < gw) ? s + rw + ov : W - 2);
  }
  for (uint32_t r = 0; r < gh; ++r) {
    uint32_t t = 3 + r * (rh + ov);
    b.y0[r] = (uint16_t)(r ? t - ov : 2);
    b.y1[r] = (uint16_t)((r + 1 < gh) ? t + rh + ov : H - 4);
  }
  return b;
}

// Tile shape of kpe_fused: NT threads = FPB frames x Wd dword columns (the multiple of 64 that
// leaves fewest idle lanes), R output rows, key tiles padded so that the lanes of a wave keep
// hitting consecutive LDS banks across the seam between two frames of the workgroup.
rmx_ctx::FusedPlan plan_fused(uint32_t W, uint32_t H) {
  rmx_ctx::FusedPlan pl;
  // opt-in (experiments): measured slower than kpe_main + kpe_finish, DESIGN.md §4.1c
  if (W % 4 != 0 || !getenv("RMX_KPE_FUSED")) return pl;
  const uint32_t Wd = W / 4;
  if (Wd > 1024) return pl;
  uint32_t best_nt = 0;
  double best_waste = 2.0;
  for (uint32_t nt = 256; nt <= 1024; nt += 64) {
    if (nt < Wd) continue;
    const uint32_t fpb = std::min(nt / Wd, 32u);
    const double waste = 1.0 - (double)(fpb * Wd) / nt;
    if (waste < best_waste - 1e-9) best_waste = waste, best_nt = nt;  // ties: the smaller workgroup
  }
  if (!best_nt) return pl;
  uint32_t R = 16;
  if (const char* e = getenv("RMX_KPE_R")) R = (uint32_t)atoi(e);
  if (R != 8 && R != 16 && R != 24) R = 16;
  for (;; R -= 8) {
    const uint32_t TR = R + 4, fpb = std::min(best_nt / Wd, 32u);
    uint32_t padw = (32 - ((TR - 1) * Wd) % 32) % 32;  // (TR * Wd + padw) == Wd (mod 32 banks)
    padw = (padw + 3) & ~3u;
    const uint32_t pitch = (TR * Wd + padw) * 4;
    const uint32_t lds = kf_lds_bytes(fpb, pitch);
    if (lds <= 160 * 1024) {
      pl.ok = true;
      pl.v16 = W % 16 == 0;
      pl.R = R;
      pl.nt = best_nt;
      pl.fpb = fpb;
      pl.pitch = pitch;
      pl.lds = lds;
      return pl;
    }
    if (R == 8) return pl;
  }
}

template <int R, bool V16>
hipError_t launch_fused(const rmx_ctx::FusedPlan& pl, const KpeFusedParams& kp, hipStream_t stream, size_t& lds_granted) {
  if (pl.lds > lds_granted) {  // dynamic LDS above the 64-KB default needs the attribute (per device: kept per context)
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&kpe_fused<R, V16>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pl.lds);
    if (e != hipSuccess) return e;
    lds_granted = pl.lds;
  }
  const uint32_t groups = (kp.n + pl.fpb - 1) / pl.fpb;
  hipLaunchKernelGGL((kpe_fused<R, V16>), dim3(groups * kp.strips), dim3(pl.nt), pl.lds, stream, kp);
  return hipGetLastError();
}

int ensure_workspace(rmx_ctx* ctx, uint32_t n) {
  if (n <= ctx->n_cap) return RMX_OK;
  if (ctx->raw) (void)hipFree(ctx->raw);
  if (ctx->kraw_base) (void)hipFree(ctx->kraw_base);
  if (ctx->tflag) (void)hipFree(ctx->tflag);
  if (ctx->cand) (void)hipFree(ctx->cand);
  if (ctx->tab_ws) (void)hipFree(ctx->tab_ws);
  ctx->tab_ws = nullptr;
  ctx->raw = nullptr;
  ctx->kraw = ctx->kraw_base = ctx->kraw_other = nullptr;
  ctx->tflag = nullptr;
  ctx->cand = nullptr;
  ctx->n_cap = 0;
  const size_t cand_bytes = ctx->fused.ok ? 256 : (size_t)n * ((ctx->H + 7) / 8) * ((ctx->W + 3) / 4) * sizeof(uint32_t);
  RMX_HIP(hipMalloc(&ctx->raw, (size_t)n * ctx->kmax * sizeof(RawDesc)));
  RMX_HIP(hipMalloc(&ctx->kraw_base, 2 * (size_t)n * sizeof(uint32_t)));
  ctx->kraw = ctx->kraw_base;
  ctx->kraw_other = ctx->kraw_base + n;
  ctx->kraw_clean = 0;
  RMX_HIP(hipMalloc(&ctx->tflag, (size_t)n * sizeof(uint32_t)));
  RMX_HIP(hipMalloc(&ctx->cand, cand_bytes));
  if (ctx->kmax > KPE_KMAX_LDS) RMX_HIP(hipMalloc(&ctx->tab_ws, (size_t)n * 2 * (((size_t)ctx->W * ctx->H + 31) / 32) * sizeof(uint32_t)));
  ctx->n_cap = n;
  ctx->ws_bytes = (size_t)n * ctx->kmax * sizeof(RawDesc) + (size_t)n * 4 + cand_bytes + kpm_workspace_bytes(ctx->kpm);
  return RMX_OK;
}

// rows of `width` bytes, `spitch` apart, packed back to back (2-byte units: descriptor rows)
__global__ __launch_bounds__(256) void pack_rows(const uint16_t* __restrict__ src, uint16_t* __restrict__ dst, size_t spitch2,
                                                 uint32_t width2, size_t total2) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total2) return;
  const size_t row = i / width2;
  dst[i] = src[row * spitch2 + (i - row * width2)];
}

__global__ __launch_bounds__(256) void copy_stream(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = src[i];
}
// A copy across PCIe done by a few workgroups instead of a DMA engine (one side is page-locked host memory, which
// the device addresses directly): a grid-stride walk, 16 bytes per lane where both sides allow it.  The grid is
// small on purpose — the waves mostly wait for the link, and the chip belongs to the kernels of the other blocks.
__global__ __launch_bounds__(256) void copy_link(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, size_t bytes) {
  const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x, nt = (size_t)gridDim.x * 256;
  if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15u) == 0) {
    const size_t n16 = bytes / 16;
    const uint4* s4 = reinterpret_cast<const uint4*>(src);
    uint4* d4 = reinterpret_cast<uint4*>(dst);
    size_t i = tid;
    for (; i + 3 * nt < n16; i += 4 * nt) {  // four loads in flight per lane
      const uint4 a = s4[i], b = s4[i + nt], c = s4[i + 2 * nt], d = s4[i + 3 * nt];
      d4[i] = a, d4[i + nt] = b, d4[i + 2 * nt] = c, d4[i + 3 * nt] = d;
    }
    for (; i < n16; i += nt) d4[i] = s4[i];
    for (size_t j = n16 * 16 + tid; j < bytes; j += nt) dst[j] = src[j];
  } else {
    for (size_t j = tid; j < bytes; j += nt) dst[j] = src[j];
  }
}

struct Timer {
  rmx_ctx* c;
  rmx_ctx::Track* t;
  std::pair<hipEvent_t, hipEvent_t>* p = nullptr;
  Timer(rmx_ctx* c_, rmx_ctx::Track* t_) : c(c_), t(t_) {
    if (!c->timing || (c->timing == 2 && t != &c->ms_kpe)) return;
    if (t->used == t->ev.size()) {
      hipEvent_t a = nullptr, b = nullptr;
      if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
      t->ev.emplace_back(a, b);
    }
    p = &t->ev[t->used++];
    (void)hipEventRecord(p->first, c->stream);
  }
  ~Timer() {
    if (p) (void)hipEventRecord(p->second, c->stream);
  }
};

void free_track(rmx_ctx::Track& t) {
  for (auto& e : t.ev) {
    (void)hipEventDestroy(e.first);
    (void)hipEventDestroy(e.second);
  }
  t.ev.clear();
  t.used = 0;
}

}  // namespace

extern "C" {

const char* rmx_version(void) { return "remap_amd 0.1 gfx950"; }

void rmx_luts(uint8_t n2o[16], uint8_t o2n[16]) {
  std::memcpy(n2o, kN2O, 16);
  std::memcpy(o2n, kO2N, 16);
}

int rmx_ctx_create(int device, uint32_t width, uint32_t height, uint32_t grid_w, uint32_t grid_h,
                   uint32_t overlap, uint32_t kmax, void* stream, rmx_ctx** out) {
  if (!out) return RMX_ERR_ARG;
  *out = nullptr;
  // Any width: a lane owns one dword of a row (the last one may be partial).  Lower bounds: the band arithmetic of
  // kpe.hpp:84-90 needs every region to start inside the frame.  x,y are uint16 in rmx_desc.
  if (width < 16 || height < 8 || width > 65532 || height > 65535) return RMX_ERR_ARG;
  if (grid_w < 1 || grid_h < 1 || grid_w > 4 || grid_h > 4 || grid_w * grid_h > 8) return RMX_ERR_ARG;
  if (kmax < 1 || kmax > KPE_KMAX_MAX) return RMX_ERR_ARG;
  {
    uint32_t rw = width / grid_w, rh = height / grid_h;
    if (rw <= overlap / 2 || rh <= overlap / 2) return RMX_ERR_ARG;
    rw -= overlap / 2;
    rh -= overlap / 2;
    if (2 + (grid_w - 1) * (rw + overlap) > width - 2 || 3 + (grid_h - 1) * (rh + overlap) > height - 4)
      return RMX_ERR_ARG;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return RMX_ERR_NOGPU;
  if (device < 0 || device >= ndev) return RMX_ERR_ARG;
  rmx_ctx* ctx = new (std::nothrow) rmx_ctx();
  if (!ctx) return RMX_ERR_DEVICE;
  ctx->device = device;
  if (stream == RMX_STREAM_OWN) {  // a non-blocking stream of the context's own (blocks in flight: include/remap/collector.hpp)
    if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess) {
      delete ctx;
      return RMX_ERR_DEVICE;
    }
    ctx->own_stream = true;
  } else {
    ctx->stream = (hipStream_t)stream;
  }
  ctx->W = width;
  ctx->H = height;
  ctx->gw = grid_w;
  ctx->gh = grid_h;
  ctx->overlap = overlap;
  ctx->kmax = kmax;
  ctx->bands = make_bands(width, height, grid_w, grid_h, overlap);
  ctx->fused = plan_fused(width, height);
  if (const char* e = getenv("RMX_SUB_MIB")) ctx->sub_mib = (uint32_t)std::max(1, atoi(e));
  if (const char* e = getenv("RMX_OVERLAP")) ctx->overlap_subs = (uint32_t)std::max(0, atoi(e));
  if (const char* e = getenv("RMX_RPS")) ctx->rps_max = (uint32_t)std::max(8, atoi(e) / 8 * 8);
  ctx->kpe_bs = width % 16 == 0;
  if (const char* e = getenv("RMX_KPE_BS")) ctx->kpe_bs = ctx->kpe_bs && atoi(e) != 0;
  if (const char* e = getenv("RMX_BS_RA")) ctx->bs_ra = ctx->bs_ra_ring = (uint32_t)std::min<int>(std::max(1, atoi(e)), (int)KPE_BS_RA_MAX);
  if (const char* e = getenv("RMX_BS_RING")) ctx->bs_ring_forced = atoi(e) != 0, ctx->bs_ring = atoi(e) != 0;
  ctx->kpe_pc = ctx->kpe_bs && width % 32 == 0 && width / 32 <= 64 && 64 % (width / 32) <= 8;
  if (const char* e = getenv("RMX_KPE_PC")) ctx->kpe_pc = ctx->kpe_pc && atoi(e) != 0;
  if (const char* e = getenv("RMX_PC_DENSE")) ctx->pc_dense_forced = atoi(e) != 0, ctx->pc_dense = atoi(e) != 0;
  if (const char* e = getenv("RMX_PC_RA")) ctx->pc_ra = ctx->pc_ra_short = (uint32_t)std::max(6, atoi(e) / 6 * 6);
  if (const char* e = getenv("RMX_PC_PLAN")) {
    unsigned a = 0, b = 0, c = 0;
    if (sscanf(e, "%u,%u,%u", &a, &b, &c) == 3 && a >= 6 && b >= 6 && c <= 100) ctx->pc_ra = a / 6 * 6, ctx->pc_ra_short = b / 6 * 6, ctx->pc_long_pct = c;
  }
  if (hipSetDevice(device) != hipSuccess || hipMalloc(&ctx->status, 64 * sizeof(uint32_t)) != hipSuccess ||
      hipMemset(ctx->status, 0, 64 * sizeof(uint32_t)) != hipSuccess ||
      hipHostMalloc(reinterpret_cast<void**>(&ctx->host_words), 16 * sizeof(uint32_t), hipHostMallocDefault) != hipSuccess) {
    if (ctx->status) (void)hipFree(ctx->status);
    if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
    delete ctx;
    return RMX_ERR_DEVICE;
  }
  *out = ctx;
  return RMX_OK;
}

int rmx_ctx_destroy(rmx_ctx* ctx) {
  if (!ctx) return RMX_ERR_ARG;
  (void)hipSetDevice(ctx->device);
  (void)hipStreamSynchronize(ctx->stream);
  if (ctx->raw) (void)hipFree(ctx->raw);
  if (ctx->tflag) (void)hipFree(ctx->tflag);
  if (ctx->kraw_base) (void)hipFree(ctx->kraw_base);
  if (ctx->cand) (void)hipFree(ctx->cand);
  if (ctx->pc_ovf) (void)hipFree(ctx->pc_ovf);
  if (ctx->tab_ws) (void)hipFree(ctx->tab_ws);
  if (ctx->status) (void)hipFree(ctx->status);
  if (ctx->host_words) (void)hipHostFree(ctx->host_words);
  if (ctx->fdf_ws) (void)hipFree(ctx->fdf_ws);
  if (ctx->pack_buf) (void)hipFree(ctx->pack_buf);
  if (ctx->turn_ev) {
    if (ctx->turns && ctx->turns->last == ctx->turn_ev) ctx->turns->last = nullptr;
    (void)hipEventDestroy(ctx->turn_ev);
  }
  for (auto e : ctx->ev_main) (void)hipEventDestroy(e);
  for (auto e : ctx->ev_fin) (void)hipEventDestroy(e);
  if (ctx->aux) (void)hipStreamDestroy(ctx->aux);
  if (ctx->blit_ws) (void)hipFree(ctx->blit_ws);
  kpm_workspace_free(ctx->kpm);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->stream);
  for (auto* t : {&ctx->ms_kpe, &ctx->ms_finish, &ctx->ms_table, &ctx->ms_kpm_build, &ctx->ms_kpm_match, &ctx->ms_blit, &ctx->ms_nic, &ctx->ms_fdf}) free_track(*t);
  delete ctx;
  return RMX_OK;
}

// status[15] of the batch whose status words just came back: keypoints of its sampled frames
static void kpe_note_density(rmx_ctx* ctx, uint32_t sampled_keypoints) {
  if (ctx->n_last == 0 || ctx->kmax > KPE_KMAX_LDS) return;  // (kpe_table_big does not sample)
  const uint64_t frames = (ctx->n_last + 31) / 32;
  // kpe_pc: the instance with eight register passes from ~250 keypoints per 320x240 frame on (measured: 4 passes
  // 0.347 / 0.37 / 0.62 / 1.73 ms per 10 000 frames at 67 / 170 / 450 / 1 100 keypoints per frame, 8 passes 0.353 / - / 0.53 / 0.85)
  if (ctx->pc_dense_forced < 0) ctx->pc_dense = (uint64_t)sampled_keypoints * 300 > frames * ctx->W * ctx->H;
  if (ctx->bs_ring_forced >= 0) return;
  // kpe_bs_ring against kpe_bs (widths kpe_pc does not take): measured crossover at 320x240 between 173 and 360 keypoints per frame
  ctx->bs_ring = (uint64_t)sampled_keypoints * 320 <= frames * ctx->W * ctx->H;
}

int rmx_kpe_take_turns(rmx_ctx* const* ctxs, uint32_t n) {
  if (!ctxs || n == 0) return RMX_ERR_ARG;
  for (uint32_t i = 0; i < n; ++i)
    if (!ctxs[i] || ctxs[i]->device != ctxs[0]->device) return RMX_ERR_ARG;
  auto turns = std::make_shared<rmx_ctx::KpeTurns>();
  for (uint32_t i = 0; i < n; ++i) {
    rmx_ctx* ctx = ctxs[i];
    RMX_HIP(hipSetDevice(ctx->device));
    if (!ctx->turn_ev) RMX_HIP(hipEventCreateWithFlags(&ctx->turn_ev, hipEventDisableTiming));
    if (ctx->turns && ctx->turns->last == ctx->turn_ev) ctx->turns->last = nullptr;
    ctx->turns = n > 1 ? turns : nullptr;
  }
  return RMX_OK;
}

int rmx_kpe_wait_turn(rmx_ctx* ctx) {
  if (!ctx) return RMX_ERR_ARG;
  if (ctx->turns && ctx->turns->last && ctx->turns->last != ctx->turn_ev) {
    RMX_HIP(hipSetDevice(ctx->device));
    RMX_HIP(hipStreamWaitEvent(ctx->stream, ctx->turns->last, 0));
  }
  return RMX_OK;
}

int rmx_sync(rmx_ctx* ctx) {
  if (!ctx) return RMX_ERR_ARG;
  RMX_HIP(hipStreamSynchronize(ctx->stream));
  return RMX_OK;
}

const char* rmx_kpe_kernel(const rmx_ctx* ctx) { return ctx ? ctx->kpe_kernel : ""; }

const char* rmx_last_error(const rmx_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }
size_t rmx_ctx_workspace_bytes(const rmx_ctx* ctx) { return ctx ? ctx->ws_bytes : 0; }

int rmx_set_timing(rmx_ctx* ctx, int enabled) {
  if (!ctx) return RMX_ERR_ARG;
  ctx->timing = enabled == 2 ? 2 : (enabled != 0);
  // (re)start a measurement window: forget the pairs recorded so far
  RMX_HIP(hipStreamSynchronize(ctx->stream));
  for (auto* t : {&ctx->ms_kpe, &ctx->ms_finish, &ctx->ms_table, &ctx->ms_kpm_build, &ctx->ms_kpm_match, &ctx->ms_blit, &ctx->ms_nic, &ctx->ms_fdf}) t->used = 0;
  return RMX_OK;
}

static int kernel_stats(rmx_ctx* ctx, const char* kernel, float* ms, uint32_t* launches);

int rmx_last_kernel_ms(rmx_ctx* ctx, const char* kernel, float* ms) { return kernel_stats(ctx, kernel, ms, nullptr); }
int rmx_kernel_launches(rmx_ctx* ctx, const char* kernel, uint32_t* launches) {
  float ms;
  return kernel_stats(ctx, kernel, &ms, launches);
}

static int kernel_stats(rmx_ctx* ctx, const char* kernel, float* ms, uint32_t* launches) {
  if (!ctx || !kernel || !ms) return RMX_ERR_ARG;
  std::string k(kernel);
  rmx_ctx::Track* t = nullptr;
  if (k == "kpe_main")
    t = &ctx->ms_kpe;
  else if (k == "kpe_finish")
    t = &ctx->ms_finish;
  else if (k == "kpe_table")
    t = &ctx->ms_table;
  else if (k == "kpm_build")
    t = &ctx->ms_kpm_build;
  else if (k == "kpm_match")
    t = &ctx->ms_kpm_match;
  else if (k == "fgm_blit")
    t = &ctx->ms_blit;
  else if (k == "nic_compress")
    t = &ctx->ms_nic;
  else if (k == "fdf")
    t = &ctx->ms_fdf;
  else
    return RMX_ERR_ARG;
  RMX_HIP(hipStreamSynchronize(ctx->stream));
  double sum = 0;
  for (size_t i = 0; i < t->used; ++i) {
    float v = 0.f;
    RMX_HIP(hipEventElapsedTime(&v, t->ev[i].first, t->ev[i].second));
    sum += v;
  }
  *ms = t->used ? (float)(sum / (double)t->used) : 0.f;
  if (launches) *launches = (uint32_t)t->used;
  return RMX_OK;
}

int rmx_malloc(rmx_ctx* ctx, size_t bytes, void** out) {
  if (!ctx || !out || bytes == 0) return RMX_ERR_ARG;
  RMX_HIP(hipSetDevice(ctx->device));
  RMX_HIP(hipMalloc(out, bytes));
  return RMX_OK;
}
int rmx_malloc_host(rmx_ctx* ctx, size_t bytes, void** out) {
  if (!ctx || !out) return RMX_ERR_ARG;
  RMX_HIP(hipSetDevice(ctx->device));
  RMX_HIP(hipHostMalloc(out, bytes, hipHostMallocDefault));
  return RMX_OK;
}
int rmx_free_host(rmx_ctx* ctx, void* ptr) {
  if (!ctx) return RMX_ERR_ARG;
  if (ptr) RMX_HIP(hipHostFree(ptr));
  return RMX_OK;
}
int rmx_free(rmx_ctx* ctx, void* ptr) {
  if (!ctx) return RMX_ERR_ARG;
  RMX_HIP(hipSetDevice(ctx->device));
  RMX_HIP(hipStreamSynchronize(ctx->stream));
  if (ptr) RMX_HIP(hipFree(ptr));
  return RMX_OK;
}
int rmx_upload(rmx_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (!ctx || !dst || !src) return RMX_ERR_ARG;
  RMX_HIP(hipSetDevice(ctx->device));
  RMX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
  return RMX_OK;
}
int rmx_download(rmx_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (!ctx || !dst || !src) return RMX_ERR_ARG;
  RMX_HIP(hipSetDevice(ctx->device));
  RMX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
  RMX_HIP(hipStreamSynchronize(ctx->stream));
  return RMX_OK;
}
int rmx_download_2d(rmx_ctx* ctx, void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t rows) {
  if (!ctx || !dst || !src || width > dpitch || width > spitch) return RMX_ERR_ARG;
  if (width == 0 || rows == 0) return RMX_OK;
  RMX_HIP(hipSetDevice(ctx->device));
  // many short rows into one tight block (a block of descriptor tables cut to the rows in use):
  // packed on the device, then ONE linear copy — a 2-D DMA of 16 384 rows of ~1 KB took 80 ms
  const uintptr_t al = reinterpret_cast<uintptr_t>(src) | spitch | width;
  if (dpitch == width && (al & 1u) == 0 && rows > 64) {
    const size_t bytes = width * rows;
    if (bytes > ctx->pack_cap) {
      RMX_HIP(hipStreamSynchronize(ctx->stream));
      if (ctx->pack_buf) (void)hipFree(ctx->pack_buf);
      ctx->pack_buf = nullptr;
      ctx->pack_cap = 0;
      RMX_HIP(hipMalloc(&ctx->pack_buf, bytes));
      ctx->pack_cap = bytes;
    }
    const size_t total2 = bytes / 2;
    hipLaunchKernelGGL(pack_rows, dim3((uint32_t)((total2 + 255) / 256)), dim3(256), 0, ctx->stream,
                       static_cast<const uint16_t*>(src), static_cast<uint16_t*>(ctx->pack_buf), spitch / 2, (uint32_t)(width / 2),
                       total2);
    RMX_HIP(hipGetLastError());
    RMX_HIP(hipMemcpyAsync(dst, ctx->pack_buf, bytes, hipMemcpyDeviceToHost, ctx->stream));
    RMX_HIP(hipStreamSynchronize(ctx->stream));
    return RMX_OK;
  }
  RMX_HIP(hipMemcpy2DAsync(dst, dpitch, src, spitch, width, rows, hipMemcpyDeviceToHost, ctx->stream));
  RMX_HIP(hipStreamSynchronize(ctx->stream));
  return RMX_OK;
}
int rmx_upload_pinned(rmx_ctx* ctx, void* dst, const void* src_pinned, size_t bytes) {
  if (!ctx || !dst || !src_pinned) return RMX_ERR_ARG;
  if (bytes == 0) return RMX_OK;
  RMX_HIP(hipSetDevice(ctx->device));
  const uint32_t blocks = (uint32_t)std::min<size_t>(256, (bytes / 16 + 255) / 256 + 1);
  hipLaunchKernelGGL(copy_link, dim3(blocks), dim3(256), 0, ctx->stream, static_cast<const uint8_t*>(src_pinned), static_cast<uint8_t*>(dst), bytes);
  RMX_HIP(hipGetLastError());
  return RMX_OK;
}
int rmx_download_pinned(rmx_ctx* ctx, void* dst_pinned, const void* src, size_t bytes) {
  if (!ctx || !dst_pinned || !src) return RMX_ERR_ARG;
  if (bytes == 0) return RMX_OK;
  RMX_HIP(hipSetDevice(ctx->device));
  const uint32_t blocks = (uint32_t)std::min<size_t>(64, (bytes / 16 + 255) / 256 + 1);
  hipLaunchKernelGGL(copy_link, dim3(blocks), dim3(256), 0, ctx->stream, static_cast<const uint8_t*>(src), static_cast<uint8_t*>(dst_pinned), bytes);
  RMX_HIP(hipGetLastError());
  return RMX_OK;
}
int rmx_download_2d_pinned(rmx_ctx* ctx, void* dst_pinned, const void* src, size_t spitch, size_t width, size_t rows) {
  if (!ctx || !dst_pinned || !src || width > spitch || ((reinterpret_cast<uintptr_t>(src) | spitch | width) & 1u)) return RMX_ERR_ARG;
  if (width == 0 || rows == 0) return RMX_OK;
  RMX_HIP(hipSetDevice(ctx->device));
  const size_t total2 = width * rows / 2;
  hipLaunchKernelGGL(pack_rows, dim3((uint32_t)((total2 + 255) / 256)), dim3(256), 0, ctx->stream, static_cast<const uint16_t*>(src),
                     static_cast<uint16_t*>(dst_pinned), spitch / 2, (uint32_t)(width / 2), total2);
  RMX_HIP(hipGetLastError());
  return RMX_OK;
}
int rmx_copy(rmx_ctx* ctx, void* dst, const void* src, size_t bytes) {
  if (!ctx || !dst || !src) return RMX_ERR_ARG;
  RMX_HIP(hipSetDevice(ctx->device));
  // big aligned copies go through a plain 16-B-per-lane streaming kernel: 6.2 TB/s on an
  // MI355X where hipMemcpyAsync device-to-device reaches 4.8 (tools/micro/copy_bw.hip) — this
  // is also the copy bench.py times as the practical HBM ceiling
  const uintptr_t a = reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src);
  if (bytes >= (1u << 20) && (a & 15u) == 0) {
    const size_t n16 = bytes / 16, tail = bytes - n16 * 16;
    hipLaunchKernelGGL(copy_stream, dim3((uint32_t)((n16 + 255) / 256)), dim3(256), 0, ctx->stream,
                       reinterpret_cast<const uint4*>(src), reinterpret_cast<uint4*>(dst), n16);
    RMX_HIP(hipGetLastError());
    if (tail)
      RMX_HIP(hipMemcpyAsync(static_cast<uint8_t*>(dst) + n16 * 16, static_cast<const uint8_t*>(src) + n16 * 16, tail,
                             hipMemcpyDeviceToDevice, ctx->stream));
    return RMX_OK;
  }
  RMX_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, ctx->stream));
  return RMX_OK;
}
int rmx_memset(rmx_ctx* ctx, void* dst, int value, size_t bytes) {
  if (!ctx || !dst) return RMX_ERR_ARG;
  RMX_HIP(hipSetDevice(ctx->device));
  RMX_HIP(hipMemsetAsync(dst, value, bytes, ctx->stream));
  return RMX_OK;
}

int rmx_kpe_batch(rmx_ctx* ctx, const uint8_t* frames, uint32_t n, uint8_t* median, rmx_desc* desc,
                  uint32_t* kcount) {
  if (!ctx || !frames || !median || !desc || !kcount || n == 0 || n > 0x0fffffffu) return RMX_ERR_ARG;
  ctx->status_cached = false;
  RMX_HIP(hipSetDevice(ctx->device));
  if (int rc = ensure_workspace(ctx, n)) return rc;
  ctx->n_last = n;
  // the append counters: the buffer the previous batch's table kernels cleared (a memset only
  // for what they did not reach: the first batch, a longer batch)
  std::swap(ctx->kraw, ctx->kraw_other);
  if (ctx->kraw_clean < n)
    RMX_HIP(hipMemsetAsync(ctx->kraw + ctx->kraw_clean, 0, (size_t)(n - ctx->kraw_clean) * sizeof(uint32_t), ctx->stream));
  const uint32_t clean_after = n;  // (of the other buffer, once this batch's table kernels have been launched)
  ctx->kraw_clean = 0;             // until then: an early error return must not leave it marked clean
  // Optional sub-batching (RMX_SUB_MIB): keeping frames + medians + candidate plane inside
  // the 256 MiB Infinity Cache between the streaming and the finishing pass was measured
  // SLOWER at every size (small launches cost more than the re-reads), so the default is
  // one launch pair for the whole batch.
  const size_t per_frame = (size_t)ctx->W * ctx->H * 2 + (size_t)((ctx->H + 7) / 8) * ((ctx->W + 3) / 4) * 4;
  uint32_t sub = (uint32_t)std::max<size_t>(1, ((size_t)ctx->sub_mib << 20) / per_frame);
  const bool overlap = ctx->overlap_subs > 1 && n >= 256 * ctx->overlap_subs;
  if (overlap) {
    sub = (n + ctx->overlap_subs - 1) / ctx->overlap_subs;
    if (!ctx->aux) RMX_HIP(hipStreamCreateWithFlags(&ctx->aux, hipStreamNonBlocking));
    while (ctx->ev_main.size() < ctx->overlap_subs) {
      hipEvent_t a, b;
      RMX_HIP(hipEventCreateWithFlags(&a, hipEventDisableTiming));
      RMX_HIP(hipEventCreateWithFlags(&b, hipEventDisableTiming));
      ctx->ev_main.push_back(a);
      ctx->ev_fin.push_back(b);
    }
  }
  if (int rc = rmx_kpe_wait_turn(ctx)) return rc;
  uint32_t sub_idx = 0;
  if (ctx->fused.ok) {
    const rmx_ctx::FusedPlan& pl = ctx->fused;
    KpeFusedParams kp{};
    kp.frames = frames;
    kp.median = median;
    kp.raw = ctx->raw;
    kp.kcount = ctx->kraw;
    kp.status = ctx->status;
    kp.n = n;
    kp.W = ctx->W;
    kp.H = ctx->H;
    kp.Wd = ctx->W / 4;
    kp.kmax = ctx->kmax;
    kp.fpb = pl.fpb;
    kp.strips = (ctx->H + pl.R - 1) / pl.R;
    kp.pitch = pl.pitch;
    hipError_t e;
    {
      Timer t(ctx, &ctx->ms_kpe);
      if (pl.R == 8)
        e = pl.v16 ? launch_fused<8, true>(pl, kp, ctx->stream, ctx->lds_fused) : launch_fused<8, false>(pl, kp, ctx->stream, ctx->lds_fused);
      else if (pl.R == 24)
        e = pl.v16 ? launch_fused<24, true>(pl, kp, ctx->stream, ctx->lds_fused) : launch_fused<24, false>(pl, kp, ctx->stream, ctx->lds_fused);
      else
        e = pl.v16 ? launch_fused<16, true>(pl, kp, ctx->stream, ctx->lds_fused) : launch_fused<16, false>(pl, kp, ctx->stream, ctx->lds_fused);
    }
    RMX_HIP(e);
    ctx->kpe_kernel = "kpe_fused";
  }
  // the two-kernel form (RMX_KPE_PAIR) and the sub-batch experiments exist for kpe_main / kpe_finish only
  const bool use_bs = !ctx->fused.ok && ctx->kpe_bs && !overlap && getenv("RMX_KPE_PAIR") == nullptr;
  const bool use_pc = use_bs && ctx->kpe_pc && (ctx->bs_ring || ctx->bs_ring_forced < 0);  // (RMX_BS_RING=0 forces kpe_bs)
  if (use_pc) {
    KpeParams p{};
    p.frames = frames;
    p.median = median;
    p.raw = ctx->raw;
    p.kcount = ctx->kraw;
    p.status = ctx->status;
    p.n = n;
    p.W = ctx->W;
    p.H = ctx->H;
    p.Wd = ctx->W / 32;  // 32-pixel chunks per row
    p.HB = 64 / p.Wd;    // image rows (of consecutive frames) per wave
    p.groups = (n + p.HB - 1) / p.HB;
    p.V = (uint64_t)n * p.Wd;
    p.rows_per_strip = std::max(6u, std::min(ctx->pc_ra, (ctx->H + 5) / 6 * 6) / 6 * 6);
    p.rows_short = std::max(6u, std::min(ctx->pc_ra_short, p.rows_per_strip) / 6 * 6);
    // long strips over the first pc_long_pct % of the rows (fewer halo rows), short ones over the rest
    p.strips_long = p.rows_short == p.rows_per_strip ? 0u : (uint32_t)((uint64_t)ctx->H * ctx->pc_long_pct / 100 / p.rows_per_strip);
    const uint32_t rest = ctx->H - p.strips_long * p.rows_per_strip;
    p.kmax = ctx->kmax;
    dim3 grid((p.groups + 3) / 4, p.strips_long + (rest + p.rows_short - 1) / p.rows_short);
    const size_t ovf_bytes = (size_t)grid.x * grid.y * 4 * PC_OVF_DWORDS * sizeof(uint32_t);
    if (ovf_bytes > ctx->pc_ovf_bytes) {  // (the kernel leaves it as it finds it: all zero)
      if (ctx->pc_ovf) {
        RMX_HIP(hipStreamSynchronize(ctx->stream));
        (void)hipFree(ctx->pc_ovf);
        ctx->pc_ovf = nullptr, ctx->pc_ovf_bytes = 0;
      }
      RMX_HIP(hipMalloc(&ctx->pc_ovf, ovf_bytes));
      RMX_HIP(hipMemsetAsync(ctx->pc_ovf, 0, ovf_bytes, ctx->stream));
      ctx->pc_ovf_bytes = ovf_bytes;
    }
    p.cand = ctx->pc_ovf;
    {
      Timer t(ctx, &ctx->ms_kpe);
      const size_t pad_lds = getenv("RMX_PC_PADLDS") ? (size_t)atoi(getenv("RMX_PC_PADLDS")) * 1024 : 0;  // occupancy sweeps: fewer workgroups per CU
      // (the dense instance keeps the patches of eight passes per roll in registers, the sparse one of four)
      ctx->kpe_kernel = ctx->pc_dense ? "kpe_pc_dense" : "kpe_pc";
      if (ctx->pc_dense) hipLaunchKernelGGL(kpe_pc<8>, grid, dim3(256), pad_lds, ctx->stream, p);
      else hipLaunchKernelGGL(kpe_pc<4>, grid, dim3(256), pad_lds, ctx->stream, p);
    }
    RMX_HIP(hipGetLastError());
  }
  if (use_bs && !use_pc) {
    KpeParams p{};
    p.frames = frames;
    p.median = median;
    p.raw = ctx->raw;
    p.kcount = ctx->kraw;
    p.status = ctx->status;
    p.n = n;
    p.W = ctx->W;
    p.H = ctx->H;
    p.Wd = ctx->W / 16;  // 16-byte chunks per row
    p.V = (uint64_t)n * p.Wd;
    p.groups = (uint32_t)((p.V + 61) / 62);
    // whole image rows per wave where that leaves few lanes idle (3 x 20 chunks at 320 pixels), else 62
    // chunks of the frames laid back to back + 2 halo lanes
    p.HB = p.Wd <= 64 && 64 % p.Wd <= 8 && !getenv("RMX_BS_HALO") ? 64 / p.Wd : 0;
    if (p.HB) p.groups = (n + p.HB - 1) / p.HB;
    const bool ring = ctx->bs_ring;
    p.rows_per_strip = std::max(6u, std::min(ring ? ctx->bs_ra_ring : ctx->bs_ra, (ctx->H + 1) / 2) / 6 * 6);
    p.kmax = ctx->kmax;
    p.cand = ctx->cand;  // (diagnostic builds only: phase stamps)
    dim3 grid((p.groups + 3) / 4, (ctx->H + 2 * p.rows_per_strip - 1) / (2 * p.rows_per_strip));
    {
      Timer t(ctx, &ctx->ms_kpe);
      const size_t pad_lds = getenv("RMX_BS_PADLDS") ? (size_t)atoi(getenv("RMX_BS_PADLDS")) * 1024 : 0;  // experiment: fewer workgroups per CU
      ctx->kpe_kernel = ring ? "kpe_bs_ring" : "kpe_bs";
      if (ring) hipLaunchKernelGGL(kpe_bs_ring, grid, dim3(256), pad_lds, ctx->stream, p);  // (keeps no candidate words)
      else hipLaunchKernelGGL(kpe_bs, grid, dim3(256), p.rows_per_strip * 256 * sizeof(uint32_t) + pad_lds, ctx->stream, p);
    }
    RMX_HIP(hipGetLastError());
  }
  for (uint32_t f0 = 0; !ctx->fused.ok && !use_bs && f0 < n; f0 += sub, ++sub_idx) {
    const uint32_t m = std::min(sub, n - f0);
    KpeParams p{};
    p.frames = frames + (size_t)f0 * ctx->W * ctx->H;
    p.median = median + (size_t)f0 * ctx->W * ctx->H;
    p.raw = ctx->raw + (size_t)f0 * ctx->kmax;
    p.kcount = ctx->kraw + f0;
    p.status = ctx->status;
    p.n = m;
    p.W = ctx->W;
    p.H = ctx->H;
    p.Wd = (ctx->W + 3) / 4;
    p.V = (uint64_t)m * p.Wd;
    p.groups = (uint32_t)((p.V + 61) / 62);
    // strips (multiples of 8 rows): enough waves to fill the chip for small batches, long
    // strips (less halo-row re-reading) for big ones
    uint32_t rps = ctx->rps_max;
    while (rps > 8 && (uint64_t)p.groups * ((ctx->H + rps - 1) / rps) < 16384) rps -= 8;
    p.rows_per_strip = rps;
    p.kmax = ctx->kmax;
    p.HB = (ctx->H + 7) / 8;
    p.cand = ctx->cand + (size_t)f0 * p.HB * p.Wd;
    dim3 grid((p.groups + 3) / 4, (ctx->H + rps - 1) / rps);
    // one kernel for both halves (kpe_mf) unless asked for the pair (experiments) or the strip is
    // longer than its LDS words hold
    const bool one_kernel = !overlap && rps <= 8 * KPE_WPS_MAX && getenv("RMX_KPE_PAIR") == nullptr;
    ctx->kpe_kernel = one_kernel ? "kpe_mf" : "kpe_main+kpe_finish";
    {
      Timer t(ctx, &ctx->ms_kpe);
      if (one_kernel) {
        if (ctx->W % 4 == 0)
          hipLaunchKernelGGL(kpe_mf<true>, grid, dim3(256), 0, ctx->stream, p);
        else
          hipLaunchKernelGGL(kpe_mf<false>, grid, dim3(256), 0, ctx->stream, p);
      } else if (ctx->W % 4 == 0) {
        hipLaunchKernelGGL(kpe_main<true>, grid, dim3(256), 0, ctx->stream, p);
      } else {
        hipLaunchKernelGGL(kpe_main<false>, grid, dim3(256), 0, ctx->stream, p);
      }
    }
    RMX_HIP(hipGetLastError());
    if (one_kernel) continue;
    const uint32_t cpf = (p.HB * p.Wd + FIN_WORDS - 1) / FIN_WORDS;
    if (overlap) {
      RMX_HIP(hipEventRecord(ctx->ev_main[sub_idx], ctx->stream));
      RMX_HIP(hipStreamWaitEvent(ctx->aux, ctx->ev_main[sub_idx], 0));
      hipLaunchKernelGGL(kpe_finish, dim3(m * cpf), dim3(256), 0, ctx->aux, p);
      RMX_HIP(hipEventRecord(ctx->ev_fin[sub_idx], ctx->aux));
    } else {
      Timer t(ctx, &ctx->ms_finish);
      hipLaunchKernelGGL(kpe_finish, dim3(m * cpf), dim3(256), 0, ctx->stream, p);
    }
    RMX_HIP(hipGetLastError());
  }
  if (overlap)
    for (uint32_t i = 0; i < sub_idx; ++i) RMX_HIP(hipStreamWaitEvent(ctx->stream, ctx->ev_fin[i], 0));
  if (ctx->turns) {
    RMX_HIP(hipEventRecord(ctx->turn_ev, ctx->stream));
    ctx->turns->last = ctx->turn_ev;
  }
  TableParams tp{};
  tp.raw = ctx->raw;
  tp.kcount = ctx->kraw;
  tp.knext = ctx->kraw_other;
  tp.desc = desc;
  tp.kout = kcount;
  tp.status = ctx->status;
  tp.n = n;
  tp.kmax = ctx->kmax;
  tp.bands = ctx->bands;
  uint32_t n2 = 1;
  while (n2 < ctx->kmax) n2 <<= 1;
  // LDS: the sort keys, or (if it fits 64 KB) the one-bit-per-pixel rank bitmap
  size_t lds = n2 * sizeof(uint64_t);
  const size_t words = ((size_t)ctx->W * ctx->H + 31) / 32;
  tp.H = ctx->H;
  tp.rank_words = 0;
  if (words * 4 <= 65536) {
    tp.rank_words = (uint32_t)words;
    lds = std::max<size_t>(128 * sizeof(uint64_t), words * 4);  // the sort then only sees frames with <= 128 keypoints
  }
  if (ctx->kmax > KPE_KMAX_LDS) {
    Timer t(ctx, &ctx->ms_table);
    hipLaunchKernelGGL(kpe_table_big, dim3(n), dim3(1024), 0, ctx->stream, tp, ctx->tab_ws, (uint32_t)words);
  } else {
    Timer t(ctx, &ctx->ms_table);
    tp.wave_ok = ctx->W <= 1024 && ctx->H <= 1024 && ctx->kmax <= 2048;
    tp.left = ctx->tflag;
    hipLaunchKernelGGL(kpe_table, dim3((n + 3) / 4), dim3(256), 0, ctx->stream, tp);
    // frames too big for the rank bitmap sort 64-bit keys in LDS: 128 KB at kmax = 16 384, above
    // the 64 KB a launch gets without the attribute
    // (the attribute belongs to the device the context runs on: remembered per context, not per process)
    if (lds > ctx->lds_table_rest) {
      RMX_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&kpe_table_rest), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      ctx->lds_table_rest = lds;
    }
    hipLaunchKernelGGL(kpe_table_rest, dim3(std::min(n, 4096u)), dim3(256), lds, ctx->stream, tp);
  }
  RMX_HIP(hipGetLastError());
  ctx->kraw_clean = clean_after;
  return RMX_OK;
}

int rmx_kpe_status(rmx_ctx* ctx, uint32_t* max_k) {
  if (!ctx) return RMX_ERR_ARG;
  RMX_HIP(hipSetDevice(ctx->device));
  uint32_t st = 0;
  if (ctx->status_cached && !max_k) {
    st = ctx->cached_status[0];
    ctx->cached_status[0] = 0;
  } else {
    RMX_HIP(hipMemcpyAsync(ctx->host_words, ctx->status, 16 * sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
    RMX_HIP(hipStreamSynchronize(ctx->stream));
    st = ctx->host_words[0];
    kpe_note_density(ctx, ctx->host_words[15]);
  }
  if (max_k) {  // diagnostic: the per-frame counts are only fetched when asked for
    *max_k = 0;
    if (ctx->n_last) {
      std::vector<uint32_t> k(ctx->n_last);
      RMX_HIP(hipMemcpy(k.data(), ctx->kraw, (size_t)ctx->n_last * 4, hipMemcpyDeviceToHost));
      for (uint32_t v : k) *max_k = std::max(*max_k, v);
    }
  }
  if (st) {
    RMX_HIP(hipMemsetAsync(ctx->status, 0, sizeof(uint32_t), ctx->stream));
    return RMX_ERR_KMAX;
  }
  return RMX_OK;
}

}  // extern "C"

#include "api_kpm_fgm.inc"
