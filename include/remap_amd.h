/*
 * remap_amd.h — C ABI of the MI355X-native alignment hot path of kataklinger/remap.
 *
 * The reference has no FFI: its boundary is the C++ template surface that main.cpp and
 * mpb.hpp compile against.  The replacement headers under include/remap/
 * keep those names and forward their heavy methods to the entry points declared here.
 * Every entry point cites the reference interface it replaces (file:line under
 * /root/reference/src).
 *
 * Conventions
 *   - plain pointers and sizes only; no C++/torch types cross this boundary;
 *   - every function returns RMX_OK (0) or a negative rmx_status; nothing throws;
 *   - all frame / table / canvas pointers are DEVICE pointers (HBM) unless the parameter
 *     is documented as host; the library never allocates or frees caller buffers;
 *   - work is enqueued on the context's HIP stream; rmx_sync() waits for it.  Functions
 *     that return host-visible scalars (counts, status words) synchronise internally
 *     and say so;
 *   - a context is thread-compatible (one thread at a time), contexts are independent.
 */
#ifndef REMAP_AMD_H
#define REMAP_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum rmx_status {
  RMX_OK = 0,
  RMX_ERR_ARG = -1,      /* null pointer, zero size, unsupported dimensions            */
  RMX_ERR_DEVICE = -2,   /* a HIP call failed; rmx_last_error() has the HIP string     */
  RMX_ERR_KMAX = -3,     /* a frame produced more keypoints than the table stride      */
  RMX_ERR_CAPACITY = -4, /* an internal workspace limit (pairs / offsets) was exceeded */
  RMX_ERR_NOGPU = -5     /* no HIP device visible                                      */
} rmx_status;

/* One keypoint = one row of a frame's descriptor table.
 * code        : kpr::code, kpr.hpp:20-23 — the 25 native nibbles of the 5x5 patch plus the
 *               weight, packed exactly as encode_keypoint does (kpe.hpp:342-379): row-major,
 *               low nibble first, except that bytes 2 and 7 hold (next row's pixel 0) in the
 *               low and (this row's pixel 4) in the high nibble; byte 12 = weight | p24 << 4.
 * region_mask : bit (grid_h*col + row) set for every kpr::grid region the point was added
 *               to (kpe.hpp:157-192,235-277; kpr.hpp:189-219).
 * x, y        : absolute image coordinates (mrl::point_t in kpe.hpp:228,303).
 * Rows of one frame are ordered x ascending, then y ascending — the order in which
 * kpe::extractor emits them (kpe.hpp:201-205,289). */
typedef struct rmx_desc {
  uint8_t code[13];
  uint8_t region_mask;
  uint16_t x;
  uint16_t y;
} rmx_desc; /* 18 bytes, alignment 2 */

/* kpm::vote, kpm.hpp:37-65 (count_ narrowed to 32 bits). */
typedef struct rmx_vote {
  int32_t dx;
  int32_t dy;
  uint32_t count;
  uint32_t valid; /* 0 = std::nullopt */
} rmx_vote;

typedef struct rmx_ctx rmx_ctx;

/* ---- context ------------------------------------------------------------------------ */

/* Replaces the construction of kpe::extractor<Grid,Overlap>{dims} (kpe.hpp:84-90) and of
 * frc::collector{dims} (frc.hpp:51-53): fixes frame width/height, the region grid
 * (4x2 overlap 16 for frc.hpp:22-26, 1x1 overlap 0 for fgs.hpp:17,85) and the stride
 * (kmax rows, 1 .. 4 194 304) of per-frame descriptor tables.  `stream` is a hipStream_t passed as
 * void* (NULL = the device's default stream; RMX_STREAM_OWN = a non-blocking stream the library creates for
 * this context and destroys with it — for callers that keep several blocks in flight without linking HIP). */
#define RMX_STREAM_OWN ((void*)(intptr_t)-1)
int rmx_ctx_create(int device, uint32_t width, uint32_t height, uint32_t grid_w,
                   uint32_t grid_h, uint32_t overlap, uint32_t kmax, void* stream,
                   rmx_ctx** out);
int rmx_ctx_destroy(rmx_ctx* ctx);
int rmx_sync(rmx_ctx* ctx);
/* Contexts of ONE device that hold blocks of a stream in flight, each on its own HIP stream: their extraction
 * kernels take turns.  rmx_kpe_batch then waits (on the device) for the extraction kernel the group launched
 * last before launching its own, and the context's table kernels and everything behind them run beside the
 * next context's extraction kernel.  Two extraction kernels side by side only share the chip; the matcher's
 * small kernels leave most of it idle.  n = 1 (or a context named in a later call) leaves the group.
 * rmx_kpe_wait_turn makes the context's stream wait for that last extraction kernel without launching one
 * (for a canvas blit that should not run beside it). */
int rmx_kpe_take_turns(rmx_ctx* const* ctxs, uint32_t n);
int rmx_kpe_wait_turn(rmx_ctx* ctx);
const char* rmx_last_error(const rmx_ctx* ctx);
/* Number of bytes of device workspace the context currently holds (diagnostics). */
size_t rmx_ctx_workspace_bytes(const rmx_ctx* ctx);

/* ---- device memory helpers -------------------------------------------------------------
 * Thin wrappers so a host program (the C++ header layer under include/remap/, a cgo/JNI
 * stub, ...) can stage buffers without linking the HIP runtime itself.  Uploads, memsets
 * and device-to-device copies are asynchronous on the context stream; rmx_download
 * synchronises. */
int rmx_malloc(rmx_ctx* ctx, size_t bytes, void** out);
int rmx_free(rmx_ctx* ctx, void* ptr);
/* Page-locked host memory: rmx_upload / rmx_download from or into it run at PCIe rate without
 * the driver's staging copy (a pageable 76.8-KB frame costs ~20 us per call, a pinned 300-MB
 * block ~12 ms in one). */
int rmx_malloc_host(rmx_ctx* ctx, size_t bytes, void** out);
int rmx_free_host(rmx_ctx* ctx, void* ptr);
int rmx_upload(rmx_ctx* ctx, void* dst_device, const void* src_host, size_t bytes);
int rmx_download(rmx_ctx* ctx, void* dst_host, const void* src_device, size_t bytes);
int rmx_copy(rmx_ctx* ctx, void* dst_device, const void* src_device, size_t bytes);
/* The same transfers done by a few workgroups of the context's stream instead of a DMA engine, for PAGE-LOCKED
 * host memory (rmx_malloc_host), which the device addresses directly.  All three are asynchronous (rmx_sync
 * waits).  For callers with blocks in flight on several contexts: DMA copies of different streams share engine
 * queues, and the few hundred kilobytes a block hands back wait behind the 300 MB the next one brings in —
 * kernel copies of different streams simply run side by side.  rmx_download_2d_pinned packs `rows` rows of
 * `width` bytes (spitch apart on the device; all even) back to back into dst. */
int rmx_upload_pinned(rmx_ctx* ctx, void* dst_device, const void* src_pinned_host, size_t bytes);
int rmx_download_pinned(rmx_ctx* ctx, void* dst_pinned_host, const void* src_device, size_t bytes);
int rmx_download_2d_pinned(rmx_ctx* ctx, void* dst_pinned_host, const void* src_device, size_t src_pitch, size_t width, size_t rows);
/* rows x width bytes from a pitched device array into a pitched host array; synchronises. */
int rmx_download_2d(rmx_ctx* ctx, void* dst_host, size_t dst_pitch, const void* src_device, size_t src_pitch,
                    size_t width, size_t rows);
int rmx_memset(rmx_ctx* ctx, void* dst_device, int value, size_t bytes);

/* ---- kpe: keypoint extraction --------------------------------------------------------- */

/* Batched kpe::extractor::extract (kpe.hpp:92-108) over n frames stored back to back
 * (row-major u8, one 4-bit palette index per byte).
 *   median : n*W*H bytes; processed pixels get ordered_to_native(4th largest of the 3x3
 *            window) (kpe.hpp:308-340), the unprocessed border is written 0 (the
 *            reference's freshly allocated image, frc.hpp:106).
 *   desc   : n*kmax rows; frame f's table starts at desc + f*kmax and holds kcount[f]
 *            rows in reference scan order.
 *   kcount : n uint32.
 * Asynchronous.  A frame with more than kmax keypoints sets the overflow flag reported by
 * rmx_kpe_status(); its table is then incomplete (kcount[f] = kmax) and must not be used. */
int rmx_kpe_batch(rmx_ctx* ctx, const uint8_t* frames, uint32_t n, uint8_t* median,
                  rmx_desc* desc, uint32_t* kcount);
/* Synchronises; returns RMX_ERR_KMAX if any frame since the last call overflowed kmax and
 * stores the largest per-frame keypoint count seen in *max_k (may be NULL). */
int rmx_kpe_status(rmx_ctx* ctx, uint32_t* max_k);

/* ---- kpm: keypoint matching ---------------------------------------------------------- */

/* Batched grid-variant kpm::match(cfg, previous, current) (kpm.hpp:395-415) with
 * frc::collector::match_config (frc.hpp:30-44: weight_switch 10, region_votes 3) for the
 * n-1 consecutive pairs (f-1, f), f = 1..n-1, of a table batch produced by rmx_kpe_batch.
 *   halo_desc/halo_kcount : optional table of the frame preceding frames[0] (the last
 *            frame of the previous batch or of the previous rank's shard); when non-NULL
 *            pair (halo, 0) is matched too, otherwise frame 0 reports matched = 0.
 *   dxdy    : n*2 int32, (dx,dy) = previous - current (kpm.hpp:96-98); (0,0) if unmatched
 *   matched : n uint8, 1 = std::optional engaged.
 * Tie-breaking reproduces libstdc++'s std::unordered_map iteration order exactly
 * (kpm.hpp:134-138) — on the device, there is no host replay. */
int rmx_kpm_seq(rmx_ctx* ctx, const rmx_desc* desc, const uint32_t* kcount, uint32_t n,
                const rmx_desc* halo_desc, const uint32_t* halo_kcount, int32_t* dxdy,
                uint8_t* matched);

/* Adaptive launch set for rmx_kpm_seq (off by default).  The matcher is a chain of kernel instances sized for
 * frames of growing density; on: the context's first step launches all of them and notes which found work, later
 * steps launch those only.  A step that needed an instance that was not launched reports flag 128 in
 * rmx_kpm_status like any other ticket left to rmx_kpm_resolve, which then redoes the step with every instance
 * (and adds what it used to the set): callers that handle flag 128 need nothing else.  Results are the same
 * either way; what changes is that idle 50-140 KB-LDS instances no longer wait for a CU while another stream's
 * extraction kernel fills the chip.  Calling it again resets what was learned. */
int rmx_kpm_adaptive(rmx_ctx* ctx, int on);
/* The instances the next rmx_kpm_seq launches: bit 0 kpm_build_small64, 1 / 2 kpm_build_medium<256> / <512>,
 * 3 the general build, 4-6 kpm_match tiers 0-2 (kpm_build_small, kpm_match_pair and kpm_declare always run). */
uint32_t rmx_kpm_launch_set(rmx_ctx* ctx);

/* Synchronises; returns RMX_ERR_CAPACITY if a matcher workspace limit (groups per frame,
 * shared codes or distinct offsets per region pair) was exceeded since the last call —
 * the affected results are then invalid.  *flags (may be NULL) receives the limit bits:
 * 128 alone = only tickets that rmx_kpm_resolve can still compute are missing;
 * RMX_KPM_REMOTE_FAILURE (after rmx_frc_finish on a sharded stream) = some rank's block overflowed a
 * table or hit a matcher limit that cannot be resolved: every rank must give the step up. */
#define RMX_KPM_REMOTE_FAILURE 0x8000u
int rmx_kpm_status(rmx_ctx* ctx, uint32_t* flags);

/* The tickets rmx_kpm_seq left to the global-memory instance of the matcher: a region of a frame
 * with more than 4 096 entries, a frame with more than 16 384 keypoints, or a (pair, region) with
 * more shared codes / distinct offsets than the LDS instances hold (rmx_kpm_status then returns RMX_ERR_CAPACITY with flag bit 128
 * and no other bit).  Computes them (same arguments as the rmx_kpm_seq call they belong to),
 * re-runs declare (kpm.hpp:199-211) and rewrites dxdy / matched; *resolved (may be NULL) =
 * number of tickets.  Positions derived from the old dxdy / matched (rmx_frc_scan) must be
 * derived again.  Synchronises; a no-op returning RMX_OK when nothing was left. */
int rmx_kpm_resolve(rmx_ctx* ctx, const rmx_desc* desc, const uint32_t* kcount, uint32_t n,
                    const rmx_desc* halo_desc, const uint32_t* halo_kcount, int32_t* dxdy,
                    uint8_t* matched, uint32_t* resolved);

/* Diagnostics: the per-region tickets (kpm::details::cast_vote, kpm.hpp:213-223; top-3
 * offsets by count, ties in unordered_map iteration order) of the last rmx_kpm_seq call.
 *   out : HOST, n*8*3 rmx_vote rows ([frame][region][rank]); valid = 0 pads short tickets.
 * Synchronises. */
int rmx_kpm_tickets(rmx_ctx* ctx, uint32_t n, rmx_vote* out);

/* frc::collector::process_frame's state update (frc.hpp:109-115,124-127) as a segmented
 * prefix sum: a matched frame accumulates its offset onto the running position, an
 * unmatched frame opens a new fragment at (0,0).
 *   first_is_start : 1 if frame 0 is the first frame of the stream (frc.hpp:83-95), else
 *                    the running state (*carry_pos, *carry_frag) of the previous batch is
 *                    continued (host pointers, updated on return).
 *   pos      : n*2 int32 frame position inside its fragment
 *   frag_id  : n uint32 fragment index (0-based, monotone)
 * Synchronises (returns the carry to the host). */
int rmx_frc_scan(rmx_ctx* ctx, const int32_t* dxdy, const uint8_t* matched, uint32_t n,
                 int first_is_start, int32_t carry_pos[2], uint32_t* carry_frag,
                 int32_t* pos, uint32_t* frag_id);

/* The same state update for a stream sharded over ranks by contiguous blocks of frames
 * (SURVEY.md §8e), without a host round trip between its steps — all three enqueue on the
 * context's stream:
 *   rmx_frc_scan_block   scans this rank's block as if the stream began there (first_is_start
 *                        = 1 on the rank that really holds frame 0) and writes the block's scan
 *                        summary {has a fragment start, x, y of its last frame relative to that
 *                        start or to the block's begin, number of starts} to `summary` (4 int32,
 *                        device) — what the ranks exchange (one 16-byte all-gather);
 *   rmx_frc_apply_blocks folds the summaries of the blocks before `rank` (world x 4 int32,
 *                        device, rank order) and moves the block into place: pos and frag_id
 *                        become what rmx_frc_scan over the whole stream gives (frc.hpp:109-115);
 *   rmx_frc_finish       synchronises once; *nfrag = fragments of the whole stream.
 * With ONE rank rmx_frc_apply_blocks may be left out: a block that is the whole stream is already in
 * place, and rmx_frc_scan_block leaves its fragment count for rmx_frc_finish.
 * Bit 2 of summary[0] tells the other ranks that this block failed for good (RMX_KPM_REMOTE_FAILURE above).
 * Bit 1 of summary[0] tells the other ranks that this block has tickets left to rmx_kpm_resolve:
 * after rmx_frc_finish, rmx_kpm_status reports flag 128 on EVERY rank if it does on one, so the
 * ranks resolve (a no-op where nothing is left) and scan again together. */
int rmx_frc_scan_block(rmx_ctx* ctx, const int32_t* dxdy, const uint8_t* matched, uint32_t n,
                       int first_is_start, int32_t* pos, uint32_t* frag_id, int32_t* summary);
int rmx_frc_apply_blocks(rmx_ctx* ctx, const int32_t* summaries, uint32_t rank, uint32_t world,
                         int32_t* pos, uint32_t* frag_id, uint32_t n);
int rmx_frc_finish(rmx_ctx* ctx, uint32_t* nfrag);

/* ---- fgm: fragment canvas ------------------------------------------------------------- */

/* fgm::fragment::blit(pos, image, packed, no) (fgm.hpp:87-97,176-188) for a batch of
 * frames into ONE canvas of `dot` histograms (fgm::dot_type = uint16[16], fgm.hpp:12-14):
 * ++canvas[y + pos.y - zero.y][x + pos.x - zero.x][frame[y][x]], uint16 wraparound.
 * Only frames whose frag_id[f] == fragment are blitted (frag_id may be NULL = all).  The
 * canvas must already cover every blit (size it with rmx_fgm_extent). */
int rmx_fgm_blit_batch(rmx_ctx* ctx, const uint8_t* frames, const int32_t* pos,
                       const uint32_t* frag_id, uint32_t fragment, uint32_t n,
                       uint16_t* canvas, uint32_t canvas_w, uint32_t canvas_h,
                       int32_t zero_x, int32_t zero_y);

/* The same for a canvas that holds no counts yet and has NOT been initialised: every dot of the canvas is
 * written (the sums of the batch, zeros where no frame lies), none is read — what a caller needs who has
 * just allocated the canvases of a block: no memset, half the canvas traffic.  n <= 65 535 frames (a
 * longer batch is cut into launches that add to each other: RMX_ERR_ARG).  No counterpart in the
 * reference, whose fragment grows its zero-initialised std::vector (fgm.hpp:190-233). */
int rmx_fgm_blit_fresh(rmx_ctx* ctx, const uint8_t* frames, const int32_t* pos, const uint32_t* frag_id,
                       uint32_t fragment, uint32_t n, uint16_t* canvas, uint32_t canvas_w, uint32_t canvas_h,
                       int32_t zero_x, int32_t zero_y);

/* fgm::fragment::ensure/extend/get_step (fgm.hpp:190-233) in closed form: the final
 * canvas of a fragment whose first blit is at (0,0) is the minimal box, grown from the
 * initial step-sized canvas in whole multiples of `step` (= frame dims) on each side, that
 * covers all blits; it does not depend on blit order.  Host-side, pure function.
 *   pos : n*2 int32 HOST positions of the fragment's frames. */
int rmx_fgm_extent(const int32_t* pos, uint32_t n, uint32_t step_w, uint32_t step_h,
                   uint32_t* canvas_w, uint32_t* canvas_h, int32_t* zero_x,
                   int32_t* zero_y);

/* fgm::fragment::blit(pos, fragment&&) canvas part (fgm.hpp:99-113): dst[y+oy][x+ox][b] +=
 * src[y][x][b] for every bin, uint16 wraparound; (ox,oy) = pos - dst.zero, already
 * inside dst. */
int rmx_fgm_merge(rmx_ctx* ctx, uint16_t* dst, uint32_t dst_w, uint32_t dst_h,
                  const uint16_t* src, uint32_t src_w, uint32_t src_h, int32_t ox,
                  int32_t oy);

/* fgm::fragment::blend (fgm.hpp:115-135): image = first arg-max bin, mask = 1 where any
 * bin is non-zero (image 0 / mask 0 elsewhere). */
int rmx_fgm_blend(rmx_ctx* ctx, const uint16_t* canvas, uint32_t canvas_w,
                  uint32_t canvas_h, uint8_t* image, uint8_t* mask);

/* ---- fgs: fragment splicing ------------------------------------------------------------ */

/* Cellular-variant kpm::match(preg, pmask, creg, cmask, cell) (kpm.hpp:371-393) for a list
 * of fragment pairs (fgs::match_partial / match_all, fgs.hpp:119-140).  Fragment i is
 * described by its single-region descriptor table (grid 1x1, overlap 0), its blend mask
 * and its canvas dimensions.
 *   tables/kcounts/masks : HOST arrays of nfrag DEVICE pointers / counts
 *   dims   : nfrag*2 uint32 HOST (width, height)
 *   pairs  : npairs*2 uint32 HOST (previous index, current index)
 *   votes  : npairs rmx_vote HOST out
 * Fragments of any size: pairs of two fragments with at most `kmax` keypoints (the context's,
 * at most 4 096) are matched out of LDS tables; larger ones, and pairs with more distinct
 * offsets than those hold, out of tables in device memory the call allocates (up to 5 967 347
 * keypoints per fragment and 2^31 point pairs per pair).  Synchronises. */
int rmx_kpm_cell_pairs(rmx_ctx* ctx, const rmx_desc* const* tables,
                       const uint32_t* kcounts, const uint8_t* const* masks,
                       const uint32_t* dims, uint32_t nfrag, const uint32_t* pairs,
                       uint32_t npairs, uint32_t cell_w, uint32_t cell_h, rmx_vote* votes);

/* ---- fdf: foreground filtering of a fragment's frames -------------------------------------- */

/* fdf::filter for n frames of ONE fragment (fdf.hpp:40-89): per frame fde::extractor::extract
 * (fde.hpp:95-120 — contours of equal median colour, cte.hpp:64-166, seeded where the frame
 * differs from the background, kept up to W*H/5 pixels), fde::mask (fde.hpp:125-146) and the
 * masked fragment::blit (fgm.hpp:71-86).
 *   bg       : background image, bw x bh bytes (device): rmx_fgm_blend of the fragment's canvas
 *   frames, medians : n images of the context's size (device)
 *   pos      : n x 2 int32 (device): frame position inside the background (position - zero)
 *   masks    : n x W x H bytes out (device): 1 = foreground
 *   ncontours: n uint32 out (device) or NULL: contours kept per frame
 *   canvas   : bh x bw x 16 uint16 (device) or NULL: receives the masked blit (added)
 * Asynchronous.  rmx_fdf_status synchronises; RMX_ERR_CAPACITY if a frame had more than 4 096
 * kept contours since the last call. */
int rmx_fdf_filter_batch(rmx_ctx* ctx, const uint8_t* bg, uint32_t bw, uint32_t bh,
                         const uint8_t* frames, const uint8_t* medians, const int32_t* pos,
                         uint32_t n, uint8_t* masks, uint32_t* ncontours, uint16_t* canvas);
int rmx_fdf_status(rmx_ctx* ctx);

/* ---- nic: the frame codec on the collect loop --------------------------------------------- */

/* nic::compress (nic.hpp:8-101; main.cpp:113-118 plugs it into frc::collector::collect as
 * Comp, called for the image and the median of every frame, frc.hpp:118) for n images of
 * `pixels` bytes each, resident on the device.  Blob i is written at out + i * stride and
 * its size to sizes[i]; rmx_nic_bound(pixels) is a stride no image can exceed.  Bytes equal
 * the reference's for every image inside its domain (runs <= 65535 pixels, literals <=
 * 16383); beyond it — where the reference's uint16 counters overflow and it corrupts its
 * heap, e.g. a blank 320x240 frame — runs and literals are split into several well-formed
 * tokens that nic::decompress decodes to the image.  Asynchronous. */
int rmx_nic_compress_batch(rmx_ctx* ctx, const uint8_t* images, uint32_t n, uint32_t pixels,
                           uint8_t* out, uint32_t stride, uint32_t* sizes);
/* Worst-case blob size of an image with `pixels` pixels (host helper, no context). */
uint32_t rmx_nic_bound(uint32_t pixels);
/* Gather the blobs contiguously: packed + offsets[i] <- out + i * stride, sizes[i] bytes
 * (offsets: n uint64 on the device, normally the exclusive scan of sizes).  Asynchronous. */
int rmx_nic_pack(rmx_ctx* ctx, const uint8_t* blobs, uint32_t stride, const uint32_t* sizes,
                 const uint64_t* offsets, uint32_t n, uint8_t* packed);
/* nic::decompress (nic.hpp:103-164) of n blobs (packed + offsets[i], sizes[i] bytes, all on
 * the device) into n images of `pixels` bytes.  A stream that is malformed or does not decode
 * to exactly `pixels` pixels is reported by rmx_nic_status.  Asynchronous. */
int rmx_nic_decompress_batch(rmx_ctx* ctx, const uint8_t* packed, const uint64_t* offsets,
                             const uint32_t* sizes, uint32_t n, uint32_t pixels, uint8_t* images);
/* Synchronises; RMX_ERR_CAPACITY if a blob did not fit its stride, RMX_ERR_ARG if a stream
 * given to rmx_nic_decompress_batch was malformed, since the last call. */
int rmx_nic_status(rmx_ctx* ctx);

/* ---- introspection ----------------------------------------------------------------------- */

/* Library version / build string, e.g. "remap_amd 0.1 gfx950". */
const char* rmx_version(void);
/* The two 16-entry luminance-rank LUTs (cpl.hpp:163-217) the kernels use. */
void rmx_luts(uint8_t native_to_ordered[16], uint8_t ordered_to_native[16]);
/* Kernel timing with HIP events recorded on the context stream around each launch (no host
 * synchronisation while recording).  rmx_set_timing(ctx, 1) opens a measurement window (and
 * drops earlier samples); rmx_last_kernel_ms synchronises the stream and returns the AVERAGE
 * duration in milliseconds of `kernel` ("kpe_main" = the extraction kernel: kpe_mf, or kpe_main of the
 * two-kernel form whose second half is "kpe_finish"; "kpe_table", "kpm_build", "kpm_match",
 * "fgm_blit", "nic_compress", "fdf") over the launches of the window (0 if none).  An event pair
 * costs ~10 us of stream time per launch on this hardware: enabled = 2 records only
 * "kpe_main" (what bench.py uses inside its timed region), 0 switches recording off. */
int rmx_set_timing(rmx_ctx* ctx, int enabled);
int rmx_last_kernel_ms(rmx_ctx* ctx, const char* kernel, float* ms);
/* Number of launches of `kernel` recorded in the current window (rmx_kpe_batch walks a big
 * batch in cache-sized sub-batches, one kpe_main + kpe_finish launch each). */
int rmx_kernel_launches(rmx_ctx* ctx, const char* kernel, uint32_t* launches);

/* Test hook: the iteration order of a libstdc++ std::unordered_map after inserting n keys whose
 * 64-bit hashes are hashes[0..n) (device) in index order — the replay the matchers use to break
 * ties like the reference (kpm.hpp:134-138,297) — as run by the global-memory instances
 * (fgs_big.hip: umap_order_big).  order[k] (device) = index of the k-th iterated key.
 * Synchronises. */
int rmx_debug_umap_order(rmx_ctx* ctx, const uint64_t* hashes, uint32_t n, uint32_t* order);

/* Diagnostic hook: copies `bytes` bytes of the keypoint extraction's scratch plane (the candidate plane of
 * kpe_main / kpe_finish; in builds with -DRMX_BS_STAMP the phase time stamps of kpe_bs's workgroups, eight
 * 64-bit words each, tools/bs_stamps.py) to host memory.  Synchronises.  No counterpart in the reference. */
/* Diagnostics: the extraction kernel the last rmx_kpe_batch of this context launched ("kpe_bs_ring", "kpe_bs",
 * "kpe_mf", ...): contexts choose by frame width and, between the two finishes of the bit-plane kernel, by the
 * keypoint density of the batch whose status words they last read. */
const char* rmx_kpe_kernel(const rmx_ctx* ctx);
int rmx_debug_kpe_scratch(rmx_ctx* ctx, void* host_dst, size_t bytes);
/* Diagnostics: the context's 64 device status words (flags, scan carry, launch-set words, and in builds with
 * -DRMX_STATS / -DRMX_KPM_STAMP the matcher's counters and phase times).  Synchronises. */
int rmx_debug_status(rmx_ctx* ctx, uint32_t* host_words64);
/* Test hook: OR `bits` into status word `word` (0 = extraction table overflow, 1 = matcher limits, sticky until a
 * status call reports them), as a kernel that hit the limit would.  Synchronises. */
int rmx_debug_poke_status(rmx_ctx* ctx, uint32_t word, uint32_t bits);

#ifdef __cplusplus
}
#endif
#endif /* REMAP_AMD_H */
