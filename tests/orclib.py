"""ctypes bindings for the CPU oracle (oracle/libremap_oracle.so) and, when built, the
reference harness libraries (oracle/_ref/*.so).  Test infrastructure: imported only by
tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

DESC_DTYPE = np.dtype([("code", np.uint8, (13,)), ("region_mask", np.uint8), ("x", np.uint16), ("y", np.uint16)],
                      align=False)
assert DESC_DTYPE.itemsize == 18
VOTE_DTYPE = np.dtype([("dx", np.int32), ("dy", np.int32), ("count", np.uint32), ("valid", np.uint32)])

u8p = C.POINTER(C.c_uint8)
u16p = C.POINTER(C.c_uint16)
u32p = C.POINTER(C.c_uint32)
i32p = C.POINTER(C.c_int32)
u64p = C.POINTER(C.c_uint64)


def ptr(a, t=C.c_void_p):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(t)


def build_oracle():
    if os.environ.get("RMX_ORACLE_SO"):  # another build of the restatement (oracle/Makefile: the sanitizer build)
        return os.environ["RMX_ORACLE_SO"]
    so = os.path.join(ORACLE_DIR, "libremap_oracle.so")
    src = os.path.join(ORACLE_DIR, "remap_oracle.cpp")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "libremap_oracle.so"])
    return so


_oracle = None


def oracle():
    global _oracle
    if _oracle is None:
        lib = C.CDLL(build_oracle())
        lib.orc_kpe.restype = C.c_uint32
        lib.orc_kpm_grid.restype = C.c_int
        lib.orc_kpm_cell.restype = C.c_int
        lib.orc_hash_code.restype = C.c_uint64
        lib.orc_hash_offset.restype = C.c_uint64
        _oracle = lib
    return _oracle


def _load_ref(name):
    so = os.path.join(ORACLE_DIR, "_ref", name)
    return C.CDLL(so) if os.path.exists(so) else None


_ref = {}


def ref_strict():
    if "s" not in _ref:
        lib = _load_ref("libref_strict.so")
        if lib is not None:
            lib.ref_kpm_grid.restype = C.c_int
            lib.ref_kpm_cell.restype = C.c_int
            lib.ref_hash_code.restype = C.c_uint64
            lib.ref_hash_offset.restype = C.c_uint64
            lib.ref_frag_new.restype = C.c_void_p
        _ref["s"] = lib
    return _ref["s"]


def ref_full():
    if "f" not in _ref:
        lib = _load_ref("libref_full.so")
        if lib is not None:
            lib.ref_kpe.restype = C.c_uint32
            lib.ref_frc.restype = C.c_uint32
            lib.ref_frc_splice.restype = C.c_uint32
        _ref["f"] = lib
    return _ref["f"]


# ---- oracle wrappers ---------------------------------------------------------------------------

def orc_luts():
    a = np.zeros(16, np.uint8)
    b = np.zeros(16, np.uint8)
    oracle().orc_luts(ptr(a), ptr(b))
    return a, b


def orc_kpe(img: np.ndarray, gw=4, gh=2, overlap=16, cap=None):
    H, W = img.shape
    img = np.ascontiguousarray(img, np.uint8)
    cap = cap or (W * H)
    med = np.zeros((H, W), np.uint8)
    desc = np.zeros(cap, DESC_DTYPE)
    k = oracle().orc_kpe(ptr(img), W, H, gw, gh, overlap, ptr(med), ptr(desc), cap)
    return med, desc[: min(k, cap)].copy(), k


def orc_kpm_grid(prev, cur, nreg=8, weight_switch=10, votes=3):
    prev = np.ascontiguousarray(prev)
    cur = np.ascontiguousarray(cur)
    d = np.zeros(2, np.int32)
    t = np.zeros(nreg * votes, VOTE_DTYPE)
    m = oracle().orc_kpm_grid(ptr(prev), len(prev), ptr(cur), len(cur), nreg, weight_switch, votes, ptr(d), ptr(t))
    return m, (int(d[0]), int(d[1])), t.reshape(nreg, votes)


def orc_kpm_cell(prev, pmask, cur, cdims, cell=(15, 15)):
    prev = np.ascontiguousarray(prev)
    cur = np.ascontiguousarray(cur)
    pmask = np.ascontiguousarray(pmask, np.uint8)
    ph, pw = pmask.shape
    ch, cw = cdims
    v = np.zeros(1, VOTE_DTYPE)
    m = oracle().orc_kpm_cell(ptr(prev), len(prev), ptr(pmask), pw, ph, ptr(cur), len(cur), cw, ch, cell[0], cell[1],
                              ptr(v))
    return m, v[0]


def orc_nic_compress(img):
    px = np.ascontiguousarray(img, np.uint8).reshape(-1)
    out = np.zeros(len(px) + 64, np.uint8)
    lib = oracle()
    lib.orc_nic_compress.restype = C.c_uint64
    n = lib.orc_nic_compress(ptr(px), C.c_uint64(len(px)), ptr(out), C.c_uint64(len(out)))
    assert n <= len(out)
    return out[:n].tobytes()


def orc_nic_decompress(pack, npix):
    b = np.frombuffer(pack, np.uint8)
    out = np.zeros(npix, np.uint8)
    lib = oracle()
    lib.orc_nic_decompress.restype = C.c_uint64
    n = lib.orc_nic_decompress(ptr(b), C.c_uint64(len(b)), ptr(out), C.c_uint64(npix))
    return out, int(n)


def ref_nic_compress(img):
    img = np.ascontiguousarray(img, np.uint8)
    H, W = img.shape
    out = np.zeros(img.size + 64, np.uint8)
    lib = ref_strict()
    lib.ref_nic_compress.restype = C.c_uint64
    n = lib.ref_nic_compress(ptr(img), W, H, ptr(out), C.c_uint64(len(out)))
    return out[:n].tobytes()


def ref_nic_decompress(pack, W, H):
    b = np.frombuffer(pack, np.uint8)
    out = np.zeros((H, W), np.uint8)
    ref_strict().ref_nic_decompress(ptr(b), C.c_uint64(len(b)), W, H, ptr(out))
    return out


def orc_umap_order(hashes):
    h = np.ascontiguousarray(hashes, np.uint64)
    o = np.zeros(len(h), np.uint32)
    oracle().orc_umap_order(ptr(h), len(h), ptr(o))
    return o


def orc_umap_order_std(hashes):
    h = np.ascontiguousarray(hashes, np.uint64)
    o = np.zeros(len(h), np.uint32)
    oracle().orc_umap_order_std(ptr(h), len(h), ptr(o))
    return o


def orc_frc(frames):
    frames = np.ascontiguousarray(frames, np.uint8)
    n, H, W = frames.shape
    matched = np.zeros(n, np.uint8)
    dxdy = np.zeros((n, 2), np.int32)
    pos = np.zeros((n, 2), np.int32)
    frag = np.zeros(n, np.uint32)
    kc = np.zeros(n, np.uint32)
    oracle().orc_frc(ptr(frames), n, W, H, ptr(matched), ptr(dxdy), ptr(pos), ptr(frag), ptr(kc))
    return dict(matched=matched, dxdy=dxdy, pos=pos, frag=frag, kcount=kc)


def orc_fgm_extent(pos, sw, sh):
    pos = np.ascontiguousarray(pos, np.int32).reshape(-1, 2)
    cw, ch = C.c_uint32(), C.c_uint32()
    zx, zy = C.c_int32(), C.c_int32()
    oracle().orc_fgm_extent(ptr(pos), len(pos), sw, sh, C.byref(cw), C.byref(ch), C.byref(zx), C.byref(zy))
    return cw.value, ch.value, zx.value, zy.value


def orc_fgm_blit(canvas, zero, img, pos):
    ch, cw, _ = canvas.shape
    H, W = img.shape
    img = np.ascontiguousarray(img, np.uint8)
    oracle().orc_fgm_blit(ptr(canvas), cw, ch, zero[0], zero[1], ptr(img), W, H, int(pos[0]), int(pos[1]))


def orc_fgm_merge(dst, src, off):
    dh, dw, _ = dst.shape
    sh, sw, _ = src.shape
    oracle().orc_fgm_merge(ptr(dst), dw, dh, ptr(np.ascontiguousarray(src)), sw, sh, int(off[0]), int(off[1]))


def orc_fgm_blend(canvas):
    ch, cw, _ = canvas.shape
    img = np.zeros((ch, cw), np.uint8)
    mask = np.zeros((ch, cw), np.uint8)
    oracle().orc_fgm_blend(ptr(np.ascontiguousarray(canvas)), cw, ch, ptr(img), ptr(mask))
    return img, mask


# ---- reference wrappers (only when oracle/_ref was built) -----------------------------------------

def ref_kpe(img, single_region=False, cap=None):
    lib = ref_full()
    H, W = img.shape
    img = np.ascontiguousarray(img, np.uint8)
    cap = cap or (W * H)
    med = np.zeros((H, W), np.uint8)
    desc = np.zeros(cap, DESC_DTYPE)
    k = lib.ref_kpe(ptr(img), W, H, int(single_region), ptr(med), ptr(desc), cap)
    assert k != 0xFFFFFFFF
    return med, desc[: min(k, cap)].copy(), k


def ref_kpm_grid(prev, cur):
    lib = ref_strict()
    prev = np.ascontiguousarray(prev)
    cur = np.ascontiguousarray(cur)
    d = np.zeros(2, np.int32)
    t = np.zeros(24, VOTE_DTYPE)
    m = lib.ref_kpm_grid(ptr(prev), len(prev), ptr(cur), len(cur), ptr(d), ptr(t))
    return m, (int(d[0]), int(d[1])), t.reshape(8, 3)


def ref_kpm_cell(prev, pmask, cur, cdims, cell=(15, 15)):
    lib = ref_strict()
    prev = np.ascontiguousarray(prev)
    cur = np.ascontiguousarray(cur)
    pmask = np.ascontiguousarray(pmask, np.uint8)
    ph, pw = pmask.shape
    ch, cw = cdims
    v = np.zeros(1, VOTE_DTYPE)
    m = lib.ref_kpm_cell(ptr(prev), len(prev), ptr(pmask), pw, ph, ptr(cur), len(cur), None, cw, ch, cell[0],
                         cell[1], ptr(v))
    return m, v[0]


def ref_frc(frames):
    lib = ref_full()
    frames = np.ascontiguousarray(frames, np.uint8)
    n, H, W = frames.shape
    frag = np.zeros(n, np.uint32)
    pos = np.zeros((n, 2), np.int32)
    kc = np.zeros(n, np.uint32)
    dims = np.zeros((n, 2), np.uint32)
    csum = np.zeros(n, np.uint64)
    nf = lib.ref_frc(ptr(frames), n, W, H, ptr(frag), ptr(pos), ptr(kc), ptr(dims), ptr(csum))
    return dict(nfrag=nf, frag=frag, pos_norm=pos, kentries=kc, dims=dims[:nf].copy(), canvas_sum=csum[:nf].copy())


def ref_frc_splice(frames):
    lib = ref_full()
    frames = np.ascontiguousarray(frames, np.uint8)
    n, H, W = frames.shape
    frag = np.zeros(n, np.uint32)
    pos = np.zeros((n, 2), np.int32)
    dims = np.zeros((n, 2), np.uint32)
    csum = np.zeros(n, np.uint64)
    ncol = C.c_uint32()
    nf = lib.ref_frc_splice(ptr(frames), n, W, H, ptr(frag), ptr(pos), ptr(dims), ptr(csum), C.byref(ncol))
    return dict(nfrag=nf, ncollected=ncol.value, frag=frag, pos=pos, dims=dims[:nf].copy(),
                canvas_sum=csum[:nf].copy())


class RefFragment:
    """fgm::fragment of the reference behind an opaque handle (strict tier)."""

    def __init__(self, step_w, step_h):
        self.lib = ref_strict()
        self.h = C.c_void_p(self.lib.ref_frag_new(step_w, step_h))

    def blit(self, pos, img, no=0):
        img = np.ascontiguousarray(img, np.uint8)
        H, W = img.shape
        self.lib.ref_frag_blit(self.h, int(pos[0]), int(pos[1]), ptr(img), W, H, C.c_uint64(no))

    def merge(self, pos, other: "RefFragment"):
        self.lib.ref_frag_merge(self.h, int(pos[0]), int(pos[1]), other.h)
        other.h = None

    def normalize(self):
        self.lib.ref_frag_normalize(self.h)

    def info(self):
        w, h = C.c_uint32(), C.c_uint32()
        zx, zy = C.c_int32(), C.c_int32()
        nf = C.c_uint64()
        self.lib.ref_frag_info(self.h, C.byref(w), C.byref(h), C.byref(zx), C.byref(zy), C.byref(nf))
        return w.value, h.value, zx.value, zy.value, nf.value

    def dots(self):
        w, h, *_ = self.info()
        out = np.zeros((h, w, 16), np.uint16)
        self.lib.ref_frag_dots(self.h, ptr(out))
        return out

    def frames(self):
        *_, nf = self.info()
        nums = np.zeros(nf, np.uint64)
        pos = np.zeros((nf, 2), np.int32)
        self.lib.ref_frag_frames(self.h, ptr(nums), ptr(pos))
        return nums, pos

    def margins(self):
        out = np.zeros(4, np.uint64)
        self.lib.ref_frag_margins(self.h, ptr(out))
        return tuple(int(v) for v in out)

    def blend(self):
        w, h, *_ = self.info()
        img = np.zeros((h, w), np.uint8)
        mask = np.zeros((h, w), np.uint8)
        self.lib.ref_frag_blend(self.h, ptr(img), ptr(mask))
        return img, mask

    def blit_packed(self, pos, img, no, blob_i, blob_m):
        img = np.ascontiguousarray(img, np.uint8)
        H, W = img.shape
        bi, bm = np.frombuffer(blob_i, np.uint8), np.frombuffer(blob_m, np.uint8)
        self.lib.ref_frag_blit_packed(self.h, int(pos[0]), int(pos[1]), ptr(img), W, H, C.c_uint64(no),
                                      ptr(bi) if len(bi) else None, C.c_uint64(len(bi)),
                                      ptr(bm) if len(bm) else None, C.c_uint64(len(bm)))

    def blobs(self):
        *_, nf = self.info()
        ni, nm = np.zeros(nf, np.uint64), np.zeros(nf, np.uint64)
        self.lib.ref_frag_blob_sizes(self.h, ptr(ni), ptr(nm))
        bi, bm = np.zeros(max(1, int(ni.sum())), np.uint8), np.zeros(max(1, int(nm.sum())), np.uint8)
        self.lib.ref_frag_blobs(self.h, ptr(bi), ptr(bm))
        oi = np.concatenate([[0], np.cumsum(ni)]).astype(int)
        om = np.concatenate([[0], np.cumsum(nm)]).astype(int)
        return [(bytes(bi[oi[k]:oi[k + 1]]), bytes(bm[om[k]:om[k + 1]])) for k in range(nf)]

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.ref_frag_free(self.h)
            self.h = None


def ref_ful_write(directory, fragments):
    """ful::write of the reference (ful.hpp:10-46) over RefFragment handles."""
    lib = ref_strict()
    arr = (C.c_void_p * len(fragments))(*[f.h for f in fragments])
    lib.ref_ful_write(str(directory).encode(), arr, len(fragments))


class RefFragmentView(RefFragment):
    """A fragment owned by a ful::read result vector (not freed on its own)."""

    def __init__(self, lib, h, owner):
        self.lib, self.h, self.owner = lib, C.c_void_p(h), owner

    def __del__(self):
        self.h = None


class _RefFulVector:
    def __init__(self, lib, h):
        self.lib, self.h = lib, C.c_void_p(h)

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.ref_ful_free(self.h)
            self.h = None


def ref_ful_read(directory):
    """ful::read of the reference (ful.hpp:48-102): list of RefFragment views."""
    lib = ref_strict()
    lib.ref_ful_read.restype = C.c_void_p
    lib.ref_ful_get.restype = C.c_void_p
    n = C.c_uint32()
    vec = _RefFulVector(lib, lib.ref_ful_read(str(directory).encode(), C.byref(n)))
    return [RefFragmentView(lib, lib.ref_ful_get(vec.h, i), vec) for i in range(n.value)]


def ref_fdf_lib():
    if "fdf" not in _ref:
        lib = _load_ref("libref_fdf.so")
        if lib is not None:
            lib.ref_fdf.restype = C.c_uint32
        _ref["fdf"] = lib
    return _ref["fdf"]


def ref_fdf(frames, want=0):
    """frc collect + fdf::filter of the reference (fdf.hpp:40-89).  Returns per-frame masks,
    kept-contour counts, fragment index; per fragment dims and canvas checksum; dots of `want`."""
    frames = np.ascontiguousarray(frames, np.uint8)
    n, H, W = frames.shape
    masks = np.zeros((n, H, W), np.uint8)
    ncont = np.zeros(n, np.uint32)
    frag = np.zeros(n, np.uint32)
    dims = np.zeros((n, 2), np.uint32)
    csum = np.zeros(n, np.uint64)
    cap = 64 * 1024 * 1024
    dots = np.zeros(cap, np.uint16)
    nf = ref_fdf_lib().ref_fdf(ptr(frames), n, W, H, ptr(masks), ptr(ncont), ptr(frag), ptr(dims), ptr(csum), want,
                               ptr(dots), C.c_uint64(cap))
    w, h = int(dims[want, 0]), int(dims[want, 1])
    return dict(nfrag=nf, masks=masks, ncontours=ncont, frag=frag, dims=dims[:nf].copy(), canvas_sum=csum[:nf].copy(),
                dots=dots[: w * h * 16].reshape(h, w, 16).copy())


def orc_fdf_mask(bg, bpos, frame, median):
    bg = np.ascontiguousarray(bg, np.uint8)
    frame = np.ascontiguousarray(frame, np.uint8)
    median = np.ascontiguousarray(median, np.uint8)
    H, W = frame.shape
    mask = np.zeros((H, W), np.uint8)
    lib = oracle()
    lib.orc_fdf_mask.restype = C.c_uint32
    k = lib.orc_fdf_mask(ptr(bg), bg.shape[1], bg.shape[0], int(bpos[0]), int(bpos[1]), ptr(frame), ptr(median), W, H,
                         ptr(mask))
    return mask, int(k)


def orc_fgm_blit_masked(canvas, zero, img, mask, pos):
    img = np.ascontiguousarray(img, np.uint8)
    mask = np.ascontiguousarray(mask, np.uint8)
    H, W = img.shape
    oracle().orc_fgm_blit_masked(ptr(canvas), canvas.shape[1], canvas.shape[0], int(zero[0]), int(zero[1]), ptr(img),
                                 ptr(mask), W, H, int(pos[0]), int(pos[1]))


def orc_fdf_filter(frames):
    """collect (oracle) + fdf::filter restated: per-frame masks, kept counts, filtered canvases."""
    frames = np.ascontiguousarray(frames, np.uint8)
    n, H, W = frames.shape
    o = orc_frc(frames)
    nfrag = int(o["frag"].max()) + 1
    masks = np.zeros((n, H, W), np.uint8)
    ncont = np.zeros(n, np.uint32)
    canvases = []
    for f in range(nfrag):
        idx = np.nonzero(o["frag"] == f)[0]
        cw, ch, zx, zy = orc_fgm_extent(o["pos"][idx], W, H)
        can = np.zeros((ch, cw, 16), np.uint16)
        for i in idx:
            orc_fgm_blit(can, (zx, zy), frames[i], o["pos"][i])
        bg, _ = orc_fgm_blend(can)
        res = np.zeros((ch, cw, 16), np.uint16)
        for i in idx:
            med, _, _ = orc_kpe(frames[i])
            p = o["pos"][i]
            masks[i], ncont[i] = orc_fdf_mask(bg, (p[0] - zx, p[1] - zy), frames[i], med)
            orc_fgm_blit_masked(res, (zx, zy), frames[i], masks[i], p)
        canvases.append(res)
    return dict(nfrag=nfrag, masks=masks, ncontours=ncont, frag=o["frag"], canvases=canvases)


def orc_fgm_margins(canvas):
    """fragment::margins (fgm.hpp:145-153,235-295) restated on a (h, w, 16) canvas: (left, top, right,
    bottom) = the fewest empty dots before the first / after the last non-empty dot of any row
    (column); a side keeps the canvas width (height) where no row (column) has a dot."""
    m = (np.asarray(canvas).reshape(canvas.shape[0], canvas.shape[1], -1) != 0).any(axis=2)
    h, w = m.shape
    rows, cols = m.any(axis=1), m.any(axis=0)
    if not rows.any():
        return (w, h, w, h)
    left = int(m[rows].argmax(axis=1).min())
    right = int(m[rows][:, ::-1].argmax(axis=1).min())
    top = int(m[:, cols].argmax(axis=0).min())
    bottom = int(m[::-1][:, cols].argmax(axis=0).min())
    return (left, top, right, bottom)
