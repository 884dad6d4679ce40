"""fgm canvas kernels on the GPU vs the CPU oracle."""
import numpy as np
import pytest

import orclib as O
from remap_amd import lib as L
from remap_amd.synth import scroller_v1

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["lds", "lds-unsorted", "bitplanes"], autouse=True)
def blit_kernel(request, monkeypatch):
    """Every test of this module runs with fgm.hip's kernel (the default; its group lists sorted by frame residue, and
    as the binning atomics leave them) and with the bit-plane kernel of fgm_bs.hip (RMX_BLIT_BS=1: frames without a
    mask, at least 64 pixels wide — the others take fgm.hip's kernels either way).  The switches are read per call."""
    monkeypatch.setenv("RMX_BLIT_BS", "1" if request.param == "bitplanes" else "0")
    monkeypatch.setenv("RMX_BLIT_STRIDE", "0" if request.param == "lds-unsorted" else "32")
    return request.param


def test_extent_matches_oracle_and_is_order_independent():
    lib = L.load()
    rng = np.random.default_rng(2)
    for _ in range(50):
        n = int(rng.integers(1, 30))
        pos = rng.integers(-700, 700, size=(n, 2)).astype(np.int32)
        pos[0] = 0
        ctx = type("C", (), {})()
        import ctypes as C
        def ext(p):
            cw, ch, zx, zy = C.c_uint32(), C.c_uint32(), C.c_int32(), C.c_int32()
            p = np.ascontiguousarray(p)
            assert lib.rmx_fgm_extent(p.ctypes.data, len(p), 320, 240, C.byref(cw), C.byref(ch), C.byref(zx),
                                      C.byref(zy)) == 0
            return cw.value, ch.value, zx.value, zy.value
        a = ext(pos)
        assert a == O.orc_fgm_extent(pos, 320, 240)
        perm = np.concatenate([[0], 1 + rng.permutation(n - 1)]).astype(int)
        assert ext(pos[perm]) == a


def test_blit_blend_merge(gpu):
    import torch

    frames, _ = scroller_v1(40, dens=60)
    n, H, W = frames.shape
    ctx = L.Context(W, H)
    rng = np.random.default_rng(4)
    pos = np.cumsum(rng.integers(-6, 7, size=(n, 2)), axis=0).astype(np.int32)
    pos[0] = 0
    cw, ch, zx, zy = ctx.fgm_extent(pos)
    assert (cw, ch, zx, zy) == O.orc_fgm_extent(pos, W, H)
    ocan = np.zeros((ch, cw, 16), np.uint16)
    for i in range(n):
        O.orc_fgm_blit(ocan, (zx, zy), frames[i], pos[i])
    d = torch.from_numpy(frames).cuda()
    canvas = torch.zeros((ch, cw, 16), dtype=torch.int16, device="cuda")
    ctx.fgm_blit(d, torch.from_numpy(pos).cuda(), canvas, (zx, zy))
    got = canvas.cpu().numpy().view(np.uint16)
    assert np.array_equal(got, ocan)
    # only one fragment id
    frag = torch.from_numpy((np.arange(n) % 2).astype(np.int32)).cuda()
    c2 = torch.zeros_like(canvas)
    ctx.fgm_blit(d, torch.from_numpy(pos).cuda(), c2, (zx, zy), frag_id=frag, fragment=1)
    o2 = np.zeros_like(ocan)
    for i in range(1, n, 2):
        O.orc_fgm_blit(o2, (zx, zy), frames[i], pos[i])
    assert np.array_equal(c2.cpu().numpy().view(np.uint16), o2)
    # blend
    img, mask = ctx.fgm_blend(canvas)
    oi, om = O.orc_fgm_blend(ocan)
    assert np.array_equal(img.cpu().numpy(), oi) and np.array_equal(mask.cpu().numpy(), om)
    # merge c2 into a bigger canvas at an offset
    big = torch.zeros((ch + 50, cw + 70, 16), dtype=torch.int16, device="cuda")
    big[5:5 + ch, 9:9 + cw] = canvas
    ob = np.zeros((ch + 50, cw + 70, 16), np.uint16)
    ob[5:5 + ch, 9:9 + cw] = ocan
    ctx.fgm_merge(big, c2, (33, 21))
    O.orc_fgm_merge(ob, o2, (33, 21))
    assert np.array_equal(big.cpu().numpy().view(np.uint16), ob)
    ctx.close()


def test_blit_uint16_wraparound(gpu):
    # > 65535 hits on one pixel/colour wrap exactly like the reference's uint16 bins (fgm.hpp:14)
    import torch

    W, H, n = 16, 8, 65536 + 5
    frame = np.arange(W * H, dtype=np.uint8).reshape(H, W) % 16
    ctx = L.Context(W, H, grid=(1, 1), overlap=0, kmax=64)
    d = torch.from_numpy(frame).cuda().expand(n, H, W).contiguous()
    pos = torch.zeros((n, 2), dtype=torch.int32, device="cuda")
    canvas = torch.zeros((H, W, 16), dtype=torch.int16, device="cuda")
    ctx.fgm_blit(d, pos, canvas, (0, 0))
    got = canvas.cpu().numpy().view(np.uint16)
    exp = np.zeros((H, W, 16), np.uint16)
    for y in range(H):
        for x in range(W):
            exp[y, x, frame[y, x]] = n % 65536
    assert np.array_equal(got, exp)
    ctx.close()


def test_blit_wide_path_wraparound_and_folds(gpu):
    # frames >= 64 pixels wide take the LDS-DMA kernel: 4-bit counters folded into bytes every 15
    # frames, bytes into the canvas every 255 — far more than either on one spot, past 65 535 too
    import torch

    W, H, n = 64, 8, 65536 + 300
    rng = np.random.default_rng(3)
    base = rng.integers(0, 16, size=(7, H, W), dtype=np.uint8)
    which = rng.integers(0, 7, size=n)
    ctx = L.Context(W, H, grid=(1, 1), overlap=0, kmax=64)
    d = torch.from_numpy(base).cuda()[torch.from_numpy(which).cuda()].contiguous()
    pos = torch.zeros((n, 2), dtype=torch.int32, device="cuda")
    canvas = torch.full((H, W, 16), 7, dtype=torch.int16, device="cuda")  # not empty: the launch ADDS
    ctx.fgm_blit(d, pos, canvas, (0, 0))
    exp = np.full((H, W, 16), 7, np.int64)
    yy, xx = np.mgrid[0:H, 0:W]
    for k in range(7):
        exp[yy, xx, base[k]] += int((which == k).sum())
    assert np.array_equal(canvas.cpu().numpy().view(np.uint16), (exp % 65536).astype(np.uint16))
    ctx.close()


def test_blit_jitter_many_hits_partial_rows(gpu):
    # hundreds of frames jittering around one place: every canvas dot is hit > 255 times, most
    # frames cover a tile's rows and columns only partly (the clamped window, the row checks)
    import torch

    rng = np.random.default_rng(12)
    for W, H, n in ((96, 40, 700), (200, 30, 400), (64, 9, 300), (96, 40, 1300)):  # the last one is binned (> 1024 frames)
        frames = rng.integers(0, 16, size=(n, H, W), dtype=np.uint8)
        pos = rng.integers(-21, 22, size=(n, 2)).astype(np.int32)
        pos[0] = 0
        ctx = L.Context(W, H, grid=(1, 1), overlap=0, kmax=64)
        cw, ch, zx, zy = ctx.fgm_extent(pos)
        ocan = np.zeros((ch, cw, 16), np.uint16)
        for i in range(n):
            O.orc_fgm_blit(ocan, (zx, zy), frames[i], pos[i])
        canvas = torch.zeros((ch, cw, 16), dtype=torch.int16, device="cuda")
        ctx.fgm_blit(torch.from_numpy(frames).cuda(), torch.from_numpy(pos).cuda(), canvas, (zx, zy))
        assert np.array_equal(canvas.cpu().numpy().view(np.uint16), ocan), (W, H)
        ctx.close()


def test_blit_narrow_frames(gpu):
    # frames under 64 pixels wide take the byte-per-lane kernel
    import torch

    rng = np.random.default_rng(13)
    for W, H, n in ((16, 8, 500), (60, 33, 300), (63, 9, 100), (16, 8, 1200)):
        frames = rng.integers(0, 16, size=(n, H, W), dtype=np.uint8)
        pos = np.cumsum(rng.integers(-3, 4, size=(n, 2)), axis=0).astype(np.int32)
        pos[0] = 0
        ctx = L.Context(W, H, grid=(1, 1), overlap=0, kmax=64)
        cw, ch, zx, zy = ctx.fgm_extent(pos)
        ocan = np.zeros((ch, cw, 16), np.uint16)
        for i in range(n):
            O.orc_fgm_blit(ocan, (zx, zy), frames[i], pos[i])
        canvas = torch.zeros((ch, cw, 16), dtype=torch.int16, device="cuda")
        ctx.fgm_blit(torch.from_numpy(frames).cuda(), torch.from_numpy(pos).cuda(), canvas, (zx, zy))
        assert np.array_equal(canvas.cpu().numpy().view(np.uint16), ocan), (W, H)
        ctx.close()


def test_blit_odd_width(gpu):
    import torch

    rng = np.random.default_rng(8)
    W, H, n = 133, 41, 30
    frames = rng.integers(0, 16, size=(n, H, W), dtype=np.uint8)
    pos = np.cumsum(rng.integers(-9, 10, size=(n, 2)), axis=0).astype(np.int32)
    pos[0] = 0
    ctx = L.Context(W, H, grid=(1, 1), overlap=0, kmax=64)
    cw, ch, zx, zy = ctx.fgm_extent(pos)
    ocan = np.zeros((ch, cw, 16), np.uint16)
    for i in range(n):
        O.orc_fgm_blit(ocan, (zx, zy), frames[i], pos[i])
    canvas = torch.zeros((ch, cw, 16), dtype=torch.int16, device="cuda")
    ctx.fgm_blit(torch.from_numpy(frames).cuda(), torch.from_numpy(pos).cuda(), canvas, (zx, zy))
    assert np.array_equal(canvas.cpu().numpy().view(np.uint16), ocan)
    ctx.close()


def test_blit_into_uninitialised_canvas(gpu):
    """rmx_fgm_blit_fresh: the canvas starts as garbage and comes out as the oracle's — dots no frame touches
    included — for wide frames (LDS-DMA kernel), narrow ones, a frame filter, and more than 255 frames on
    one tile (a tile's second fold must add to its first)."""
    import torch

    rng = np.random.default_rng(12)
    for (W, H, n, spread) in [(320, 240, 40, 400), (32, 16, 30, 60), (128, 64, 700, 6)]:
        frames = rng.integers(0, 16, size=(n, H, W), dtype=np.uint8)
        pos = np.cumsum(rng.integers(-spread, spread + 1, size=(n, 2)), axis=0).astype(np.int32)
        pos[0] = 0
        keep = (rng.random(n) < 0.7).astype(np.uint32)
        keep[0] = 1
        ctx = L.Context(W, H, grid=(1, 1), overlap=0, kmax=64)
        cw, ch, zx, zy = ctx.fgm_extent(pos)
        want = np.zeros((ch, cw, 16), np.uint16)
        for i in range(n):
            if keep[i]:
                O.orc_fgm_blit(want, (zx, zy), frames[i], pos[i])
        canvas = torch.full((ch, cw, 16), 0x5555, dtype=torch.int16, device="cuda")
        ctx.fgm_blit(torch.from_numpy(frames).cuda(), torch.from_numpy(pos).cuda(), canvas, (zx, zy),
                     frag_id=torch.from_numpy(keep).cuda(), fragment=1, fresh=True)
        assert np.array_equal(canvas.cpu().numpy().view(np.uint16), want), (W, H)
        ctx.close()


def test_sharded_blit_sums_to_the_whole_canvas(gpu):
    # two "ranks" blit their halves into sub-canvases; adding them at their offsets inside the
    # global box (uint16 wraparound) must give the single-rank canvas
    import torch
    from remap_amd.pipeline import Aligner

    frames, _ = scroller_v1(60, dens=80)
    frames = frames.copy()
    frames[25] = 6  # a fragment break inside the first shard
    al = Aligner(320, 240)
    d = torch.from_numpy(frames).cuda()
    out = al.align(d, blit=True)
    whole = {k: (v[0].clone(), v[1], v[2]) for k, v in out["canvases"].items()}
    parts = []
    for lo, hi in ((0, 31), (31, 60)):
        al._bufs = {k: v for k, v in al._bufs.items() if not (isinstance(k, tuple) and k[0] == "canvas")}
        parts.append({k: (v[0].clone(), v[1], v[2]) for k, v in al.blit(d[lo:hi], out["pos"], out["frag"], lo, hi).items()})
    for fid, (canvas, zero, (gw, gh, gz)) in whole.items():
        assert tuple(canvas.shape[:2]) == (gh, gw) and zero == gz
        acc = torch.zeros_like(canvas)
        for part in parts:
            if fid in part:
                c, z, g = part[fid]
                assert g == (gw, gh, gz)
                al.ctx.fgm_merge(acc, c, (z[0] - gz[0], z[1] - gz[1]))
        assert torch.equal(acc, canvas), fid
    al.close()


def test_blit_more_than_65535_frames_in_one_call(gpu):
    """A launch accumulates into 16-bit halves, so rmx_fgm_blit_batch cuts a batch of more than
    65 535 frames into several launches; counts beyond 65 535 wrap like fgm::dot_type."""
    import torch

    n, H, W = 70000, 8, 16
    rng = np.random.default_rng(4)
    frames = rng.integers(0, 16, size=(n, H, W), dtype=np.uint8)
    frames[:, 0, 0] = 5  # one bin that sees every frame: 70 000 mod 65 536
    pos = np.zeros((n, 2), np.int32)
    pos[n // 2:, 0] = 3  # second half shifted: two overlapping placements
    ctx = L.Context(W, H, grid=(1, 1), overlap=0, kmax=64)
    cw, ch, zx, zy = ctx.fgm_extent(pos)
    canvas = torch.zeros((ch, cw, 16), dtype=torch.int16, device="cuda")
    ctx.fgm_blit(torch.from_numpy(frames).cuda(), torch.from_numpy(pos).cuda(), canvas, (zx, zy))
    got = canvas.cpu().numpy().view(np.uint16)
    want = np.zeros((ch, cw, 16), np.uint64)
    for half, x0 in ((frames[: n // 2], 0), (frames[n // 2:], 3)):
        for c in range(16):
            want[:H, x0 - zx:x0 - zx + W, c] += (half == c).sum(axis=0).astype(np.uint64)
    assert np.array_equal(got, (want % 65536).astype(np.uint16))
    assert int(got[0, 0, 5]) == (n // 2) % 65536 and int(want[0, 0, 5]) == n // 2
    ctx.close()
