"""ctypes binding of the C ABI (include/remap_amd.h) + build helper.

The shared library is built in-tree (remap_amd/libremap_amd.so) by ``build()`` with one hipcc
invocation for gfx950.  There is NO CPU fallback: if the library is missing or no GPU is
visible, every compute entry point raises.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SO_PATH = os.path.join(HERE, "libremap_amd.so")
SRC_DIR = os.path.join(HERE, "csrc")

DESC_DTYPE = np.dtype([("code", np.uint8, (13,)), ("region_mask", np.uint8), ("x", np.uint16), ("y", np.uint16)],
                      align=False)
VOTE_DTYPE = np.dtype([("dx", np.int32), ("dy", np.int32), ("count", np.uint32), ("valid", np.uint32)])
DESC_BYTES = 18

RMX_OK = 0
RMX_ERR_ARG, RMX_ERR_DEVICE, RMX_ERR_KMAX, RMX_ERR_CAPACITY, RMX_ERR_NOGPU = -1, -2, -3, -4, -5
RMX_KPM_REMOTE_FAILURE = 0x8000  # rmx_kpm_status flag after rmx_frc_finish: some rank's block failed for good
STATUS = {0: "RMX_OK", -1: "RMX_ERR_ARG", -2: "RMX_ERR_DEVICE", -3: "RMX_ERR_KMAX", -4: "RMX_ERR_CAPACITY",
          -5: "RMX_ERR_NOGPU"}

# every symbol include/remap_amd.h declares
SYMBOLS = [
    "rmx_ctx_create", "rmx_ctx_destroy", "rmx_sync", "rmx_last_error", "rmx_ctx_workspace_bytes",
    "rmx_kpe_batch", "rmx_kpe_status", "rmx_kpm_seq", "rmx_kpm_status", "rmx_kpm_resolve", "rmx_kpm_tickets", "rmx_frc_scan", "rmx_frc_scan_block", "rmx_frc_apply_blocks", "rmx_frc_finish", "rmx_fgm_blit_batch", "rmx_fgm_extent",
    "rmx_fgm_merge", "rmx_fgm_blend", "rmx_kpm_cell_pairs", "rmx_version", "rmx_luts", "rmx_set_timing",
    "rmx_last_kernel_ms", "rmx_kernel_launches", "rmx_malloc", "rmx_free", "rmx_upload", "rmx_download", "rmx_download_2d", "rmx_copy", "rmx_upload_pinned", "rmx_download_pinned", "rmx_download_2d_pinned",
    "rmx_memset", "rmx_malloc_host", "rmx_free_host", "rmx_nic_compress_batch", "rmx_nic_bound", "rmx_nic_pack", "rmx_nic_status", "rmx_nic_decompress_batch", "rmx_fdf_filter_batch", "rmx_fdf_status",
    "rmx_debug_umap_order",
    "rmx_debug_kpe_scratch", "rmx_debug_status", "rmx_debug_poke_status", "rmx_kpe_kernel",
    "rmx_fgm_blit_fresh",
    "rmx_kpm_adaptive", "rmx_kpm_launch_set", "rmx_kpe_take_turns", "rmx_kpe_wait_turn",
]


def kpe_take_turns(contexts):
    """The extraction kernels of these contexts (one device, a stream each) take turns: rmx_kpe_take_turns."""
    arr = (C.c_void_p * len(contexts))(*[getattr(c.h, "value", c.h) for c in contexts])
    rc = contexts[0].lib.rmx_kpe_take_turns(arr, len(contexts))
    if rc != RMX_OK:
        raise RemapError(rc, "rmx_kpe_take_turns")


class RemapError(RuntimeError):
    def __init__(self, code, where, detail=""):
        self.code = code
        super().__init__(f"{where}: {STATUS.get(code, code)} {detail}".strip())


def sources():
    return sorted(os.path.join(SRC_DIR, f) for f in os.listdir(SRC_DIR)
                  if f.endswith((".hip", ".h", ".inc")))


def needs_build():
    if not os.path.exists(SO_PATH):
        return True
    t = os.path.getmtime(SO_PATH)
    deps = sources() + [os.path.join(ROOT, "include", "remap_amd.h")]
    return any(os.path.getmtime(s) > t for s in deps)


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -> remap_amd/libremap_amd.so (cross-compiles without a GPU)."""
    if not force and not needs_build():
        return SO_PATH
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC"]
    cmd += os.environ.get("RMX_CFLAGS", "").split()  # experiment switches (tools/), never set by tests/bench
    cmd += [os.path.join(SRC_DIR, "api.hip"), "-o", SO_PATH]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return SO_PATH


_lib = None


def load():
    """Load the library (building it if the sources are newer).  Raises if it cannot."""
    global _lib
    if _lib is not None:
        return _lib
    if needs_build():
        build()
    # torch bundles its own libamdhip64 (same SONAME as /opt/rocm's).  Import it first so that
    # this library binds to the HIP runtime torch already loaded; two runtimes in one
    # process do not share devices, streams or allocations.
    import torch  # noqa: F401

    # RMX_SO: another build of the library for A/B timings on one box (tools/), never set by tests/bench
    lib = C.CDLL(os.environ.get("RMX_SO") or SO_PATH)
    vp, u32, i32 = C.c_void_p, C.c_uint32, C.c_int32
    lib.rmx_version.restype = C.c_char_p
    lib.rmx_last_error.restype = C.c_char_p
    lib.rmx_last_error.argtypes = [vp]
    lib.rmx_ctx_workspace_bytes.restype = C.c_size_t
    lib.rmx_ctx_workspace_bytes.argtypes = [vp]
    lib.rmx_ctx_create.argtypes = [C.c_int, u32, u32, u32, u32, u32, u32, vp, C.POINTER(vp)]
    lib.rmx_ctx_destroy.argtypes = [vp]
    lib.rmx_sync.argtypes = [vp]
    lib.rmx_kpe_take_turns.argtypes = [C.POINTER(vp), u32]
    lib.rmx_kpe_wait_turn.argtypes = [vp]
    lib.rmx_kpe_batch.argtypes = [vp, vp, u32, vp, vp, vp]
    lib.rmx_kpe_status.argtypes = [vp, C.POINTER(u32)]
    lib.rmx_kpm_seq.argtypes = [vp, vp, vp, u32, vp, vp, vp, vp]
    lib.rmx_kpm_status.argtypes = [vp, C.POINTER(u32)]
    lib.rmx_kpm_adaptive.argtypes = [vp, C.c_int]
    lib.rmx_kpm_launch_set.argtypes = [vp]
    lib.rmx_kpm_launch_set.restype = u32
    lib.rmx_kpm_resolve.argtypes = [vp, vp, vp, u32, vp, vp, vp, vp, C.POINTER(u32)]
    lib.rmx_debug_umap_order.argtypes = [vp, vp, u32, vp]
    lib.rmx_debug_kpe_scratch.argtypes = [vp, vp, C.c_size_t]
    lib.rmx_debug_status.argtypes = [vp, vp]
    lib.rmx_debug_poke_status.argtypes = [vp, u32, u32]
    lib.rmx_kpe_kernel.argtypes = [vp]
    lib.rmx_kpe_kernel.restype = C.c_char_p
    lib.rmx_kpm_tickets.argtypes = [vp, u32, vp]
    lib.rmx_frc_scan.argtypes = [vp, vp, vp, u32, C.c_int, C.POINTER(i32), C.POINTER(u32), vp, vp]
    lib.rmx_frc_scan_block.argtypes = [vp, vp, vp, u32, C.c_int, vp, vp, vp]
    lib.rmx_frc_apply_blocks.argtypes = [vp, vp, u32, u32, vp, vp, u32]
    lib.rmx_frc_finish.argtypes = [vp, C.POINTER(u32)]
    lib.rmx_fgm_blit_batch.argtypes = [vp, vp, vp, vp, u32, u32, vp, u32, u32, i32, i32]
    lib.rmx_fgm_blit_fresh.argtypes = [vp, vp, vp, vp, u32, u32, vp, u32, u32, i32, i32]
    lib.rmx_fgm_extent.argtypes = [vp, u32, u32, u32, C.POINTER(u32), C.POINTER(u32), C.POINTER(i32),
                                   C.POINTER(i32)]
    lib.rmx_fgm_merge.argtypes = [vp, vp, u32, u32, vp, u32, u32, i32, i32]
    lib.rmx_fgm_blend.argtypes = [vp, vp, u32, u32, vp, vp]
    lib.rmx_kpm_cell_pairs.argtypes = [vp, vp, vp, vp, vp, u32, vp, u32, u32, u32, vp]
    lib.rmx_luts.argtypes = [vp, vp]
    lib.rmx_nic_compress_batch.argtypes = [vp, vp, u32, u32, vp, u32, vp]
    lib.rmx_nic_bound.argtypes = [u32]
    lib.rmx_nic_bound.restype = u32
    lib.rmx_nic_pack.argtypes = [vp, vp, u32, vp, vp, u32, vp]
    lib.rmx_nic_status.argtypes = [vp]
    lib.rmx_fdf_filter_batch.argtypes = [vp, vp, u32, u32, vp, vp, vp, u32, vp, vp, vp]
    lib.rmx_fdf_status.argtypes = [vp]
    lib.rmx_nic_decompress_batch.argtypes = [vp, vp, vp, vp, u32, u32, vp]
    sz = C.c_size_t
    lib.rmx_malloc.argtypes = [vp, sz, C.POINTER(vp)]
    lib.rmx_free.argtypes = [vp, vp]
    lib.rmx_malloc_host.argtypes = [vp, sz, C.POINTER(vp)]
    lib.rmx_free_host.argtypes = [vp, vp]
    lib.rmx_upload.argtypes = [vp, vp, vp, sz]
    lib.rmx_download.argtypes = [vp, vp, vp, sz]
    lib.rmx_download_2d.argtypes = [vp, vp, sz, vp, sz, sz, sz]
    lib.rmx_copy.argtypes = [vp, vp, vp, sz]
    lib.rmx_upload_pinned.argtypes = [vp, vp, vp, sz]
    lib.rmx_download_pinned.argtypes = [vp, vp, vp, sz]
    lib.rmx_download_2d_pinned.argtypes = [vp, vp, vp, sz, sz, sz]
    lib.rmx_memset.argtypes = [vp, vp, C.c_int, sz]
    lib.rmx_version.argtypes = []
    lib.rmx_set_timing.argtypes = [vp, C.c_int]
    lib.rmx_last_kernel_ms.argtypes = [vp, C.c_char_p, C.POINTER(C.c_float)]
    lib.rmx_kernel_launches.argtypes = [vp, C.c_char_p, C.POINTER(u32)]
    _lib = lib
    return lib


def luts():
    lib = load()
    a = np.zeros(16, np.uint8)
    b = np.zeros(16, np.uint8)
    lib.rmx_luts(a.ctypes.data, b.ctypes.data)
    return a, b


class Context:
    """rmx_ctx: one per (device, frame size, grid).  Buffers are torch CUDA tensors whose
    ``data_ptr()`` crosses the C ABI; torch is plumbing for device memory and streams only."""

    def __init__(self, width, height, grid=(4, 2), overlap=16, kmax=2048, device=0, stream=None):
        import torch

        self.lib = load()
        self.torch = torch
        if not torch.cuda.is_available():
            raise RemapError(-5, "Context", "no HIP device visible (the HIP path has no CPU fallback)")
        self.device = torch.device("cuda", device)
        self.W, self.H, self.grid, self.overlap, self.kmax = width, height, grid, overlap, kmax
        self.nreg = grid[0] * grid[1]
        if stream is None:
            stream = torch.cuda.current_stream(self.device).cuda_stream
        self.stream = stream
        h = C.c_void_p()
        rc = self.lib.rmx_ctx_create(device, width, height, grid[0], grid[1], overlap, kmax, C.c_void_p(stream),
                                     C.byref(h))
        if rc:
            raise RemapError(rc, "rmx_ctx_create")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.lib.rmx_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, where):
        if rc:
            raise RemapError(rc, where, self.lib.rmx_last_error(self.h).decode())

    def sync(self):
        self._chk(self.lib.rmx_sync(self.h), "rmx_sync")

    def copy(self, dst, src):
        """rmx_copy: device-to-device copy of a contiguous tensor on the context stream."""
        assert dst.is_contiguous() and src.is_contiguous() and dst.numel() * dst.element_size() == src.numel() * src.element_size()
        self._chk(self.lib.rmx_copy(self.h, dst.data_ptr(), src.data_ptr(), src.numel() * src.element_size()), "rmx_copy")

    def set_timing(self, on=True):
        """True / 1: HIP event pairs around every kernel; 2: around kpe_main only; False: off."""
        self.lib.rmx_set_timing(self.h, int(on))

    def kernel_ms(self, name):
        v = C.c_float()
        self._chk(self.lib.rmx_last_kernel_ms(self.h, name.encode(), C.byref(v)), "rmx_last_kernel_ms")
        return v.value

    def kernel_launches(self, name):
        v = C.c_uint32()
        self._chk(self.lib.rmx_kernel_launches(self.h, name.encode(), C.byref(v)), "rmx_kernel_launches")
        return v.value

    def workspace_bytes(self):
        return self.lib.rmx_ctx_workspace_bytes(self.h)

    # ---- kpe -------------------------------------------------------------------------------
    def alloc_tables(self, n):
        t = self.torch
        desc = t.empty((n, self.kmax, DESC_BYTES), dtype=t.uint8, device=self.device)
        kcount = t.empty(n, dtype=t.int32, device=self.device)
        return desc, kcount

    def kpe(self, frames, median=None, desc=None, kcount=None):
        """frames: (n, H, W) uint8 CUDA tensor.  Returns (median, desc, kcount) device tensors."""
        t = self.torch
        assert frames.is_cuda and frames.dtype == t.uint8 and frames.is_contiguous()
        n = frames.shape[0]
        assert tuple(frames.shape[1:]) == (self.H, self.W)
        if median is None:
            median = t.empty_like(frames)
        if desc is None:
            desc, kcount = self.alloc_tables(n)
        self._chk(self.lib.rmx_kpe_batch(self.h, frames.data_ptr(), n, median.data_ptr(), desc.data_ptr(),
                                         kcount.data_ptr()), "rmx_kpe_batch")
        return median, desc, kcount

    def kpe_status(self, want_max=True):
        if not want_max:
            return self.lib.rmx_kpe_status(self.h, None), 0
        mk = C.c_uint32()
        rc = self.lib.rmx_kpe_status(self.h, C.byref(mk))
        return rc, mk.value

    # ---- kpm -------------------------------------------------------------------------------
    def kpm_seq(self, desc, kcount, halo=None):
        t = self.torch
        n = desc.shape[0]
        dxdy = t.empty((n, 2), dtype=t.int32, device=self.device)
        matched = t.empty(n, dtype=t.uint8, device=self.device)
        hd, hk = (halo[0].data_ptr(), halo[1].data_ptr()) if halo is not None else (None, None)
        self._chk(self.lib.rmx_kpm_seq(self.h, desc.data_ptr(), kcount.data_ptr(), n, hd, hk, dxdy.data_ptr(),
                                       matched.data_ptr()), "rmx_kpm_seq")
        return dxdy, matched

    def kpm_resolve(self, desc, kcount, dxdy, matched, halo=None):
        """Tickets the LDS instances left to the global-memory matcher (kpm_status flag 128):
        computed, declare re-run, dxdy / matched rewritten in place.  Returns their number."""
        n = desc.shape[0]
        hd, hk = (halo[0].data_ptr(), halo[1].data_ptr()) if halo is not None else (None, None)
        res = C.c_uint32()
        self._chk(self.lib.rmx_kpm_resolve(self.h, desc.data_ptr(), kcount.data_ptr(), n, hd, hk, dxdy.data_ptr(),
                                           matched.data_ptr(), C.byref(res)), "rmx_kpm_resolve")
        return res.value

    def debug_umap_order(self, hashes):
        """iteration order of a libstdc++ unordered_map (test hook): int64 CUDA tensor of hashes -> int32 order"""
        order = self.torch.empty(hashes.shape[0], dtype=self.torch.int32, device=self.device)
        self._chk(self.lib.rmx_debug_umap_order(self.h, hashes.data_ptr(), hashes.shape[0], order.data_ptr()), "rmx_debug_umap_order")
        return order

    def kpm_adaptive(self, on=True):
        """rmx_kpm_seq launches only the matcher instances this context's stream has needed so far (a step that
        needs another reports flag 128 and kpm_resolve redoes it): for callers that handle flag 128."""
        self._chk(self.lib.rmx_kpm_adaptive(self.h, 1 if on else 0), "rmx_kpm_adaptive")

    def kpe_wait_turn(self):
        """This context's stream waits for the extraction kernel its group launched last (kpe_take_turns)."""
        self._chk(self.lib.rmx_kpe_wait_turn(self.h), "rmx_kpe_wait_turn")

    def kpe_kernel(self):
        """Name of the extraction kernel the last kpe() launched."""
        return self.lib.rmx_kpe_kernel(self.h).decode()

    def kpm_launch_set(self):
        return int(self.lib.rmx_kpm_launch_set(self.h))

    def kpm_status(self):
        fl = C.c_uint32()
        rc = self.lib.rmx_kpm_status(self.h, C.byref(fl))
        return rc, fl.value

    def kpm_tickets(self, n):
        out = np.zeros((n, 8, 3), VOTE_DTYPE)
        self._chk(self.lib.rmx_kpm_tickets(self.h, n, out.ctypes.data), "rmx_kpm_tickets")
        return out

    def kpm_cell_pairs(self, tables, kcounts, masks, dims, pairs, cell=(15, 15)):
        """Cellular kpm::match for fragment pairs.  tables/masks: lists of CUDA tensors,
        kcounts: list of ints, dims: [(w, h)], pairs: [(previous, current)].  Returns VOTE_DTYPE array."""
        nf, npairs = len(tables), len(pairs)
        tp = (C.c_void_p * nf)(*[t.data_ptr() for t in tables])
        mp = (C.c_void_p * nf)(*[m.data_ptr() for m in masks])
        kc = np.ascontiguousarray(kcounts, np.uint32)
        dm = np.ascontiguousarray(dims, np.uint32).reshape(-1, 2)
        pr = np.ascontiguousarray(pairs, np.uint32).reshape(-1, 2)
        votes = np.zeros(npairs, VOTE_DTYPE)
        if npairs:
            self._chk(self.lib.rmx_kpm_cell_pairs(self.h, tp, kc.ctypes.data, mp, dm.ctypes.data, nf, pr.ctypes.data,
                                                  npairs, cell[0], cell[1], votes.ctypes.data), "rmx_kpm_cell_pairs")
        return votes

    def frc_scan(self, dxdy, matched, first_is_start=True, carry=None):
        t = self.torch
        n = dxdy.shape[0]
        pos = t.empty((n, 2), dtype=t.int32, device=self.device)
        frag = t.empty(n, dtype=t.int32, device=self.device)
        cp = (C.c_int32 * 2)(*(carry[0] if carry else (0, 0)))
        cf = C.c_uint32(carry[1] if carry else 0)
        self._chk(self.lib.rmx_frc_scan(self.h, dxdy.data_ptr(), matched.data_ptr(), n, int(first_is_start), cp,
                                        C.byref(cf), pos.data_ptr(), frag.data_ptr()), "rmx_frc_scan")
        return pos, frag, ((cp[0], cp[1]), cf.value)

    def frc_scan_block(self, dxdy, matched, first_is_start):
        """This rank's block scanned on its own (no synchronisation): (pos, frag, summary) device
        tensors; summary = the 4 int32 the ranks exchange."""
        t = self.torch
        n = dxdy.shape[0]
        pos = t.empty((n, 2), dtype=t.int32, device=self.device)
        frag = t.empty(n, dtype=t.int32, device=self.device)
        summ = t.empty((1, 4), dtype=t.int32, device=self.device)
        self._chk(self.lib.rmx_frc_scan_block(self.h, dxdy.data_ptr(), matched.data_ptr(), n, int(first_is_start),
                                              pos.data_ptr(), frag.data_ptr(), summ.data_ptr()), "rmx_frc_scan_block")
        return pos, frag, summ

    def frc_apply_blocks(self, summaries, rank, world, pos, frag):
        """Fold the summaries of the blocks before `rank` into pos / frag (in place, no synchronisation)."""
        assert summaries.is_cuda and summaries.dtype == self.torch.int32 and summaries.is_contiguous()
        assert summaries.numel() == 4 * world
        self._chk(self.lib.rmx_frc_apply_blocks(self.h, summaries.data_ptr(), rank, world, pos.data_ptr(), frag.data_ptr(),
                                                pos.shape[0]), "rmx_frc_apply_blocks")

    def frc_finish(self):
        """Synchronise; fragments of the whole stream (also caches the status words)."""
        nf = C.c_uint32(0)
        self._chk(self.lib.rmx_frc_finish(self.h, C.byref(nf)), "rmx_frc_finish")
        return nf.value

    # ---- fdf -------------------------------------------------------------------------------
    def fdf_filter(self, bg, frames, medians, pos, canvas=None, check=True):
        """fdf::filter for the frames of one fragment.  bg: (bh, bw) uint8 CUDA tensor (blend of
        the fragment); frames/medians: (n, H, W) uint8; pos: (n, 2) int32 positions inside the
        background; canvas: optional (bh, bw, 16) int16 tensor receiving the masked blit.
        Returns (masks (n, H, W) uint8, ncontours (n,) int32) device tensors."""
        t = self.torch
        n = frames.shape[0]
        assert bg.is_contiguous() and frames.is_contiguous() and medians.is_contiguous() and pos.is_contiguous()
        assert pos.dtype == t.int32 and tuple(frames.shape[1:]) == (self.H, self.W)
        masks = t.empty_like(frames)
        ncont = t.empty(n, dtype=t.int32, device=self.device)
        if canvas is not None:
            assert tuple(canvas.shape) == (bg.shape[0], bg.shape[1], 16) and canvas.is_contiguous()
        self._chk(self.lib.rmx_fdf_filter_batch(self.h, bg.data_ptr(), bg.shape[1], bg.shape[0], frames.data_ptr(),
                                                medians.data_ptr(), pos.data_ptr(), n, masks.data_ptr(),
                                                ncont.data_ptr(), canvas.data_ptr() if canvas is not None else None),
                  "rmx_fdf_filter_batch")
        if check:
            self._chk(self.lib.rmx_fdf_status(self.h), "rmx_fdf_status")
        return masks, ncont

    # ---- nic -------------------------------------------------------------------------------
    def nic_compress(self, images, pack=True):
        """nic::compress of every image of `images` ((n, ...) uint8 CUDA tensor, one image per
        leading index).  pack=True: (packed uint8 CUDA tensor, sizes int64 host array) with
        blob i at packed[offsets[i]:offsets[i] + sizes[i]], offsets = exclusive cumsum;
        pack=False: (blobs (n, stride) CUDA tensor, sizes CUDA tensor)."""
        t = self.torch
        assert images.is_cuda and images.dtype == t.uint8 and images.is_contiguous()
        n = images.shape[0]
        pixels = images[0].numel()
        stride = int(self.lib.rmx_nic_bound(pixels))
        blobs = t.empty((n, stride), dtype=t.uint8, device=self.device)
        sizes = t.empty(n, dtype=t.int32, device=self.device)
        self._chk(self.lib.rmx_nic_compress_batch(self.h, images.data_ptr(), n, pixels, blobs.data_ptr(), stride,
                                                  sizes.data_ptr()), "rmx_nic_compress_batch")
        if not pack:
            return blobs, sizes
        self._chk(self.lib.rmx_nic_status(self.h), "rmx_nic_status")
        sz = sizes.cpu().numpy().astype(np.int64)
        offs = np.concatenate([[0], np.cumsum(sz)[:-1]]).astype(np.uint64)
        packed = t.empty(int(sz.sum()), dtype=t.uint8, device=self.device)
        doffs = t.from_numpy(offs.view(np.int64)).to(self.device)
        self._chk(self.lib.rmx_nic_pack(self.h, blobs.data_ptr(), stride, sizes.data_ptr(), doffs.data_ptr(), n,
                                        packed.data_ptr()), "rmx_nic_pack")
        self.sync()
        return packed, sz

    def nic_decompress(self, packed, sizes, pixels):
        """nic::decompress of len(sizes) blobs laid back to back in `packed` (uint8 CUDA tensor);
        sizes: host int array.  Returns (n, pixels) uint8 CUDA tensor."""
        t = self.torch
        sz = np.ascontiguousarray(sizes, np.int64)
        n = len(sz)
        offs = np.concatenate([[0], np.cumsum(sz)[:-1]]).astype(np.int64)
        doffs = t.from_numpy(offs).to(self.device)
        dsz = t.from_numpy(sz.astype(np.int32)).to(self.device)
        out = t.empty((n, pixels), dtype=t.uint8, device=self.device)
        self._chk(self.lib.rmx_nic_decompress_batch(self.h, packed.data_ptr(), doffs.data_ptr(), dsz.data_ptr(), n,
                                                    pixels, out.data_ptr()), "rmx_nic_decompress_batch")
        self._chk(self.lib.rmx_nic_status(self.h), "rmx_nic_status")
        return out

    # ---- fgm -------------------------------------------------------------------------------
    def fgm_extent(self, pos_host):
        pos_host = np.ascontiguousarray(pos_host, np.int32).reshape(-1, 2)
        cw, ch = C.c_uint32(), C.c_uint32()
        zx, zy = C.c_int32(), C.c_int32()
        rc = self.lib.rmx_fgm_extent(pos_host.ctypes.data, len(pos_host), self.W, self.H, C.byref(cw), C.byref(ch),
                                     C.byref(zx), C.byref(zy))
        self._chk(rc, "rmx_fgm_extent")
        return cw.value, ch.value, zx.value, zy.value

    def fgm_blit(self, frames, pos, canvas, zero, frag_id=None, fragment=0, fresh=False):
        """fresh: the canvas is uninitialised memory and receives every dot (rmx_fgm_blit_fresh)"""
        ch, cw = canvas.shape[0], canvas.shape[1]
        n = frames.shape[0]
        fn = self.lib.rmx_fgm_blit_fresh if fresh else self.lib.rmx_fgm_blit_batch
        self._chk(fn(self.h, frames.data_ptr(), pos.data_ptr(), frag_id.data_ptr() if frag_id is not None else None, fragment, n,
                     canvas.data_ptr(), cw, ch, zero[0], zero[1]), "rmx_fgm_blit_fresh" if fresh else "rmx_fgm_blit_batch")

    def fgm_merge(self, dst, src, off):
        self._chk(self.lib.rmx_fgm_merge(self.h, dst.data_ptr(), dst.shape[1], dst.shape[0], src.data_ptr(),
                                         src.shape[1], src.shape[0], off[0], off[1]), "rmx_fgm_merge")

    def fgm_blend(self, canvas):
        t = self.torch
        ch, cw = canvas.shape[0], canvas.shape[1]
        img = t.empty((ch, cw), dtype=t.uint8, device=self.device)
        mask = t.empty((ch, cw), dtype=t.uint8, device=self.device)
        self._chk(self.lib.rmx_fgm_blend(self.h, canvas.data_ptr(), cw, ch, img.data_ptr(), mask.data_ptr()),
                  "rmx_fgm_blend")
        return img, mask


def desc_to_numpy(desc, kcount):
    """Device tables -> list of structured numpy arrays (one per frame)."""
    d = desc.cpu().numpy()
    k = kcount.cpu().numpy()
    return [d[i, : k[i]].reshape(-1).view(DESC_DTYPE).copy() for i in range(len(k))]
