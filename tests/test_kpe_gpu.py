"""kpe parity on the GPU: HIP kernels (through the C ABI) vs the CPU oracle, bit-exact."""
import numpy as np
import pytest

import orclib as O
from remap_amd import lib as L
from remap_amd.synth import scroller_v1, upscale2

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["bs", "bs-halo", "ring", "ring-halo", "pc", "pc-short", "pc-dense", "mf"])
def kpe_path(request, monkeypatch):
    """The extraction kernels a context may pick (read from the environment when it is created): kpe_bs
    (3x3 pass on bit planes, widths that are multiples of 16; whole image rows per wave, or 62 chunks + 2
    halo lanes with short strips), kpe_bs_ring (the same streaming pass with the finish rolled into the row loop,
    which a context otherwise picks by the density of the batch before), kpe_pc (kpe_pc.hip: candidates by pairwise
    comparison, medians patched in the LDS ring; takes the ring kernel's place where a frame's rows fill a wave — other
    widths fall through to kpe_bs_ring — with strips of 30 and of 6 rows, and its instance for dense streams, which
    keeps the patches of eight passes per roll in registers instead of four) and kpe_mf (any width)."""
    if request.param == "mf":
        monkeypatch.setenv("RMX_KPE_BS", "0")
    else:
        monkeypatch.setenv("RMX_BS_RING", "0" if request.param.startswith("bs") else "1")
        monkeypatch.setenv("RMX_KPE_PC", "1" if request.param.startswith("pc") else "0")
    if request.param == "pc-short":
        monkeypatch.setenv("RMX_PC_RA", "6")
    if request.param.startswith("pc"):
        monkeypatch.setenv("RMX_PC_DENSE", "1" if request.param == "pc-dense" else "0")
    if request.param.endswith("-halo"):
        monkeypatch.setenv("RMX_BS_HALO", "1")
        monkeypatch.setenv("RMX_BS_RA", "6")
    return request.param


def run_kpe(frames, grid=(4, 2), overlap=16, kmax=4096):
    import torch

    n, H, W = frames.shape
    ctx = L.Context(W, H, grid=grid, overlap=overlap, kmax=kmax)
    d = torch.from_numpy(frames).cuda()
    med, desc, kc = ctx.kpe(d)
    rc, maxk = ctx.kpe_status()
    assert rc == 0, (rc, maxk)
    tabs = L.desc_to_numpy(desc, kc)
    med = med.cpu().numpy()
    ctx.close()
    return med, tabs


def check_against_oracle(frames, grid=(4, 2), overlap=16, kmax=4096, idx=None):
    med, tabs = run_kpe(frames, grid, overlap, kmax)
    idx = range(len(frames)) if idx is None else idx
    for i in idx:
        omed, odesc, k = O.orc_kpe(frames[i], grid[0], grid[1], overlap)
        assert np.array_equal(med[i], omed), f"median differs, frame {i}"
        assert len(tabs[i]) == k, f"keypoint count frame {i}: {len(tabs[i])} vs {k}"
        assert tabs[i].tobytes() == odesc.tobytes(), f"descriptor table differs, frame {i}"


@pytest.mark.parametrize("dens", [200, 40])
def test_kpe_scroller_320x240(gpu, dens, kpe_path):
    frames, _ = scroller_v1(24, dens=dens)
    check_against_oracle(frames)


def test_kpe_random_noise(gpu, kpe_path):
    # every pixel a candidate: stresses the queue, the 5x5 path and large K
    rng = np.random.default_rng(3)
    frames = rng.integers(0, 16, size=(3, 64, 128), dtype=np.uint8)
    check_against_oracle(frames, kmax=16384)


def test_kpe_flat_and_edges(gpu, kpe_path):
    frames = np.zeros((4, 48, 160), np.uint8)
    frames[1] = 7
    frames[2, :, 80:] = 9           # vertical edge
    frames[3, 24:, :] = 3           # horizontal edge
    frames[3, 10:14, 10:14] = 12
    check_against_oracle(frames)


def test_kpe_ragged_width_and_single_region(gpu):
    # widths that do not divide into 62-lane groups; grid<1,1>, overlap 0 (fgs.hpp:17,85)
    rng = np.random.default_rng(5)
    for (H, W) in [(40, 132), (57, 260), (33, 500)]:
        base = rng.integers(0, 16, size=(H // 8 + 1, W // 8 + 1), dtype=np.uint8)
        img = np.kron(base, np.ones((8, 8), np.uint8))[:H, :W]
        frames = np.stack([img, np.roll(img, 3, axis=1), np.roll(img, 5, axis=0)])
        check_against_oracle(np.ascontiguousarray(frames), grid=(1, 1), overlap=0)


def test_kpe_640x480_upscaled(gpu, kpe_path):
    frames, _ = scroller_v1(4, dens=100)
    check_against_oracle(upscale2(frames), idx=[0, 3])


def test_kpe_kmax_overflow_is_reported(gpu, kpe_path):
    """More keypoints than table rows: RMX_ERR_KMAX with the count that is needed (every extraction kernel
    counts all of a frame's keypoints, whatever it could store), and a context with that many rows takes them."""
    import torch

    rng = np.random.default_rng(9)
    frames = rng.integers(0, 16, size=(2, 64, 128), dtype=np.uint8)
    need = max(O.orc_kpe(f)[2] for f in frames)
    ctx = L.Context(128, 64, kmax=64)
    ctx.kpe(torch.from_numpy(frames).cuda())
    rc, maxk = ctx.kpe_status()
    assert rc == -3 and maxk == need
    ctx.close()
    ctx = L.Context(128, 64, kmax=need)
    med, desc, kc = ctx.kpe(torch.from_numpy(frames).cuda())
    assert ctx.kpe_status()[0] == 0 and int(kc.max()) == need
    ctx.close()


def test_kpe_many_frames_matches_single(gpu, kpe_path):
    # batch result == per-frame result (frames packed across waves must not interact)
    frames, _ = scroller_v1(40, dens=60)
    med, tabs = run_kpe(frames)
    for i in (0, 17, 39):
        m1, t1 = run_kpe(frames[i:i + 1])
        assert np.array_equal(med[i], m1[0])
        assert tabs[i].tobytes() == t1[0].tobytes()


def test_kpe_widths_not_multiple_of_four(gpu):
    # cropped windows have arbitrary widths (main.cpp crops the feed to the action window)
    rng = np.random.default_rng(17)
    for (H, W, grid, ov) in [(41, 133, (1, 1), 0), (60, 322, (4, 2), 16), (48, 161, (4, 2), 16), (37, 23, (1, 1), 0),
                             (64, 130, (4, 2), 16), (50, 127, (4, 2), 16)]:
        base = rng.integers(0, 16, size=(H // 5 + 1, W // 5 + 1), dtype=np.uint8)
        img = np.kron(base, np.ones((5, 5), np.uint8))[:H, :W]
        noise = rng.integers(0, 16, size=(H, W), dtype=np.uint8)
        frames = np.ascontiguousarray(np.stack([img, np.roll(img, 2, axis=1), noise, np.roll(img, 3, axis=0)]))
        check_against_oracle(frames, grid=grid, overlap=ov, kmax=16384)


def test_kpe_large_frame_sort_path_kmax_16384(gpu, kpe_path):
    """A frame above 524 288 pixels has no rank bitmap: kpe_table_rest sorts 64-bit keys in LDS,
    128 KB of dynamic LDS at kmax = 16 384 (the default of kpe::extractor in the drop-in layer)
    — needs the launch attribute; more than 8 192 keypoints so the sort really is that wide."""
    rng = np.random.default_rng(23)
    H, W = 768, 1024
    img = np.full((H, W), 6, np.uint8)
    m = rng.random((H, W)) < 0.15
    img[m] = rng.integers(0, 16, int(m.sum()))
    k = O.orc_kpe(img, 1, 1, 0)[2]
    assert 8192 < k <= 16384, k
    check_against_oracle(img[None], grid=(1, 1), overlap=0, kmax=16384)
    check_against_oracle(img[None], grid=(4, 2), overlap=16, kmax=16384)


def test_kpe_fused_kernel_matches_oracle(gpu, monkeypatch):
    """The single-kernel variant (kpe_fused.hip, opt-in through RMX_KPE_FUSED: measured slower
    than kpe_main + kpe_finish) gives the same tables: widths that are and are
    not multiples of 16, partial last strips, several frames per workgroup, dense frames that
    need more than one pass over the candidate queue."""
    monkeypatch.setenv("RMX_KPE_FUSED", "1")
    frames, _ = scroller_v1(9, dens=40)
    check_against_oracle(frames)
    rng = np.random.default_rng(29)
    for (H, W, grid, ov) in [(41, 132, (1, 1), 0), (60, 324, (4, 2), 16), (100, 64, (1, 1), 0), (250, 1284, (4, 2), 16)]:
        base = rng.integers(0, 16, size=(H // 5 + 1, W // 5 + 1), dtype=np.uint8)
        img = np.kron(base, np.ones((5, 5), np.uint8))[:H, :W]
        small = H * W < 20000
        third = rng.integers(0, 16, size=(H, W), dtype=np.uint8) if small else np.roll(img, 7, axis=1)
        frames = np.ascontiguousarray(np.stack([img, np.roll(img, 2, axis=1), third, np.roll(img, 3, axis=0), img]))
        check_against_oracle(frames, grid=grid, overlap=ov, kmax=16384)


def test_kpe_tables_beyond_16384_rows(gpu):
    """contexts with kmax above 16 384 (blended canvases of large fragments): kpe_table_big ranks
    the keypoints through a bitmap in global memory — a noise canvas with ~60 000 keypoints"""
    rng = np.random.default_rng(31)
    H, W = 300, 420
    img = rng.integers(0, 16, size=(1, H, W), dtype=np.uint8)
    k = O.orc_kpe(img[0], 1, 1, 0)[2]
    assert k > 16384
    check_against_oracle(img, grid=(1, 1), overlap=0, kmax=1 << 17)
    two = np.ascontiguousarray(np.stack([img[0], np.roll(img[0], 5, axis=1)]))
    check_against_oracle(two, grid=(1, 1), overlap=0, kmax=1 << 17)


def test_kpe_bs_shapes(gpu, kpe_path):
    """Widths that are multiples of 16 in every lane layout of kpe_bs (whole rows per wave: 16 ... 64 chunks
    dividing 64 with few lanes idle; otherwise 62 chunks + halo lanes), heights that are not multiples of
    the strip pairs, single rows of chunks, batches that are not multiples of the rows per wave — and a
    noise frame whose candidates need several rounds of the hand-out list."""
    rng = np.random.default_rng(41)
    for (H, W, grid, ov, n) in [(37, 16, (1, 1), 0, 5), (50, 32, (1, 1), 0, 7), (61, 64, (1, 1), 0, 4), (48, 160, (4, 2), 16, 5),
                                (45, 336, (4, 2), 16, 4), (70, 512, (4, 2), 16, 3), (66, 1040, (4, 2), 16, 2), (240, 320, (4, 2), 16, 7)]:
        base = rng.integers(0, 16, size=(n, H // 4 + 1, W // 4 + 1), dtype=np.uint8)
        frames = np.ascontiguousarray(np.kron(base, np.ones((1, 4, 4), np.uint8))[:, :H, :W])
        frames[n - 1] = rng.integers(0, 16, size=(H, W), dtype=np.uint8)  # noise: every pixel a candidate
        check_against_oracle(frames, grid=grid, overlap=ov, kmax=1 << 17)


def test_kpe_paths_agree_on_a_long_batch(gpu, monkeypatch):
    """kpe_bs, kpe_bs_ring, kpe_mf and both instances of kpe_pc on 3 001 frames: medians, counts and tables byte for
    byte (the batch is not a multiple of the rows a wave holds; dense frames so every workgroup hands out > 2 048
    candidates, every wave of the ring kernels appends its records in several batches and kpe_pc's rolls overflow
    the registers of its sparse instance); and a context left to itself runs kpe_pc, its dense instance after a
    dense batch, the sparse one again after a sparse batch — same bytes either way."""
    import torch

    frames, _ = scroller_v1(3001, dens=40)
    d = torch.from_numpy(frames).cuda()
    res = []
    names = ("kpe_bs", "kpe_bs_ring", "kpe_mf", "kpe_pc", "kpe_pc_dense")
    for (bs, ring, pc, dense), name in zip((("1", "0", "0", "0"), ("1", "1", "0", "0"), ("0", "0", "0", "0"), ("1", "1", "1", "0"), ("1", "1", "1", "1")), names):
        monkeypatch.setenv("RMX_KPE_BS", bs)
        monkeypatch.setenv("RMX_BS_RING", ring)
        monkeypatch.setenv("RMX_KPE_PC", pc)
        monkeypatch.setenv("RMX_PC_DENSE", dense)
        ctx = L.Context(320, 240, kmax=4096)
        med, desc, kc = ctx.kpe(d)
        assert ctx.kpe_status()[0] == 0 and ctx.kpe_kernel() == name
        res.append((med.clone(), desc.clone(), kc.clone()))
        ctx.close()
    k = res[0][2].cpu().numpy()
    a = res[0][1].cpu().numpy()
    for other, name in zip(res[1:], names[1:]):
        assert torch.equal(res[0][0], other[0]), f"medians differ between kpe_bs and {name}"
        assert torch.equal(res[0][2], other[2]), "keypoint counts differ"
        b = other[1].cpu().numpy()
        assert all(a[i, : k[i]].tobytes() == b[i, : k[i]].tobytes() for i in range(len(k))), f"tables differ ({name})"
    # the choice by density (kpe_table samples every 32nd frame's count; read back with the status words)
    monkeypatch.delenv("RMX_BS_RING")
    monkeypatch.delenv("RMX_KPE_PC")
    monkeypatch.delenv("RMX_PC_DENSE")
    monkeypatch.setenv("RMX_KPE_BS", "1")
    monkeypatch.setenv("RMX_TIMING", "1")
    sparse, _ = scroller_v1(640, dens=200)
    ds = torch.from_numpy(sparse).cuda()
    ctx = L.Context(320, 240, kmax=4096)
    ref_sparse = None
    for blk, want, kernel in ((ds, None, "kpe_pc"), (ds, None, "kpe_pc"), (d[:640], None, "kpe_pc"), (d[:640], res[0], "kpe_pc_dense"),
                              (ds, None, "kpe_pc_dense"), (ds, None, "kpe_pc")):
        med, desc, kc = ctx.kpe(blk)
        assert ctx.kpe_kernel() == kernel
        assert ctx.kpe_status()[0] == 0  # (brings the density word back: the NEXT batch's kernel is chosen by it)
        if blk is ds:
            if ref_sparse is None:
                ref_sparse = (med.clone(), kc.clone(), desc.clone())
            else:
                assert torch.equal(med, ref_sparse[0]) and torch.equal(kc, ref_sparse[1])
                kk = kc.cpu().numpy()
                x, y = desc.cpu().numpy(), ref_sparse[2].cpu().numpy()
                assert all(x[i, : kk[i]].tobytes() == y[i, : kk[i]].tobytes() for i in range(len(kk)))
        elif want is not None:
            assert torch.equal(med, want[0][:640]) and torch.equal(kc, want[2][:640])
    ctx.close()
