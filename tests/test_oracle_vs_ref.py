"""Randomised comparison of the CPU oracle with the reference's own code (oracle/_ref,
built by oracle/build_ref.sh).  Skipped where the reference libraries are absent (they
do not exist on a checkout without /root/reference; the golden vectors cover that case)."""
import numpy as np
import pytest

import orclib as O
from remap_amd.synth import scroller_v1

pytestmark = pytest.mark.skipif(O.ref_strict() is None or O.ref_full() is None,
                                reason="oracle/_ref not built (needs /root/reference)")


def structured(rng, H, W, block):
    base = rng.integers(0, 16, size=(H // block + 1, W // block + 1), dtype=np.uint8)
    img = np.kron(base, np.ones((block, block), np.uint8))[:H, :W].copy()
    for _ in range(H * W // 100):
        img[rng.integers(0, H), rng.integers(0, W)] = rng.integers(0, 16)
    return img


def test_kpe_random_images():
    rng = np.random.default_rng(100)
    for (H, W, b) in [(30, 112, 4), (64, 200, 7), (100, 324, 13), (240, 320, 3), (55, 128, 2)]:
        img = structured(rng, H, W, b)
        for single in (False, True):
            a = O.orc_kpe(img, *((1, 1, 0) if single else (4, 2, 16)))
            r = O.ref_kpe(img, single)
            assert a[2] == r[2] and np.array_equal(a[0], r[0]) and a[1].tobytes() == r[1].tobytes(), (H, W, single)


def test_kpm_grid_random_pairs():
    rng = np.random.default_rng(101)
    imgs = [structured(rng, 240, 320, b) for b in (5, 5, 9)]
    tabs = [O.orc_kpe(i)[1] for i in imgs]
    tabs.append(O.orc_kpe(np.roll(imgs[0], 3, axis=1))[1])
    tabs.append(O.orc_kpe(np.roll(imgs[0], (2, -1), axis=(0, 1)))[1])
    for a in range(len(tabs)):
        for b in range(len(tabs)):
            x, y = O.orc_kpm_grid(tabs[a], tabs[b]), O.ref_kpm_grid(tabs[a], tabs[b])
            assert x[0] == y[0] and x[1] == y[1] and x[2].tobytes() == y[2].tobytes(), (a, b)


def test_frc_stream_with_breaks():
    fr, _ = scroller_v1(50, dens=60)
    fr = fr.copy()
    fr[17] = 6
    o, r = O.orc_frc(fr), O.ref_frc(fr)
    assert np.array_equal(o["frag"], r["frag"])
    for f in range(r["nfrag"]):
        sel = o["frag"] == f
        cw, ch, zx, zy = O.orc_fgm_extent(o["pos"][sel], 320, 240)
        assert [cw, ch] == r["dims"][f].tolist()
        assert np.array_equal(o["pos"][sel] - [zx, zy], r["pos_norm"][sel])


def test_fragment_random_blits():
    rng = np.random.default_rng(102)
    fr, _ = scroller_v1(12, width=64, height=32, dens=30)
    f = O.RefFragment(64, 32)
    pos = np.cumsum(rng.integers(-40, 41, size=(12, 2)), axis=0).astype(np.int32)
    pos[0] = 0
    for i in range(12):
        f.blit(pos[i], fr[i], i)
        assert O.orc_fgm_extent(pos[: i + 1], 64, 32) == f.info()[:4]
    cw, ch, zx, zy = f.info()[:4]
    can = np.zeros((ch, cw, 16), np.uint16)
    for i in range(12):
        O.orc_fgm_blit(can, (zx, zy), fr[i], pos[i])
    assert np.array_equal(can, f.dots())
    a, b = O.orc_fgm_blend(can), f.blend()
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert O.orc_fgm_margins(can) == f.margins()


def test_fragment_margins_equal_reference():
    """fragment::margins (fgm.hpp:145-153,235-295) on canvases with empty borders of every kind: the
    restatement the GPU test checks the drop-in layer against must agree with the reference itself"""
    rng = np.random.default_rng(103)
    assert O.RefFragment(16, 8).margins() == O.orc_fgm_margins(np.zeros((8, 16, 16), np.uint16))  # nothing blitted
    for case in range(12):
        f = O.RefFragment(32, 16)
        blank = np.zeros((16, 32), np.uint8)
        n = int(rng.integers(1, 6))
        pos = rng.integers(-70, 71, size=(n, 2)).astype(np.int32)
        for i in range(n):
            f.blit(pos[i], blank, i)  # (bin 0 counts too: every blitted dot is non-empty)
        cw, ch, zx, zy = f.info()[:4]
        can = np.zeros((ch, cw, 16), np.uint16)
        for i in range(n):
            O.orc_fgm_blit(can, (zx, zy), blank, pos[i])
        assert np.array_equal(can, f.dots())
        assert O.orc_fgm_margins(can) == f.margins(), (case, pos.tolist())


def test_reference_reads_our_fragment_files(tmp_path):
    """ful::read of the reference parses files written by remap_amd.ful.pack (and vice versa)."""
    from remap_amd import ful

    rng = np.random.default_rng(9)
    want = []
    for k in range(3):
        h, w = int(rng.integers(1, 5)) * 8, int(rng.integers(1, 4)) * 16
        dots = rng.integers(0, 65536, size=(h, w, 16)).astype(np.uint16)
        zero = (int(rng.integers(-40, 1)), int(rng.integers(-20, 1)))
        frames = [(int(rng.integers(0, 1 << 40)), (int(rng.integers(-99, 99)), int(rng.integers(-99, 99))),
                   rng.integers(0, 256, size=int(rng.integers(0, 20)), dtype=np.uint8).tobytes(),
                   rng.integers(0, 256, size=int(rng.integers(0, 20)), dtype=np.uint8).tobytes())
                  for _ in range(int(rng.integers(0, 6)))]
        (tmp_path / str(k)).write_bytes(ful.pack(dots, zero, frames))
        want.append((dots, zero, frames))
    back = O.ref_ful_read(tmp_path)
    assert len(back) == 3
    for f, (dots, zero, frames) in zip(back, want):
        w, h, zx, zy, nf = f.info()
        assert (h, w) == dots.shape[:2] and (zx, zy) == zero and nf == len(frames)
        assert np.array_equal(f.dots(), dots)
        nums, pos = f.frames()
        assert nums.tolist() == [fr[0] for fr in frames] and pos.tolist() == [list(fr[1]) for fr in frames]
        assert f.blobs() == [(fr[2], fr[3]) for fr in frames]
    # the reference writes them back byte-identically
    out = tmp_path / "again"
    out.mkdir()
    O.ref_ful_write(out, back)
    for k in range(3):
        assert (out / str(k)).read_bytes() == (tmp_path / str(k)).read_bytes()


def test_nic_random_images_equal_reference():
    rng = np.random.default_rng(31)
    for k in range(300):
        h, w = int(rng.integers(1, 40)), int(rng.integers(1, 60))
        p = rng.random() ** 2
        img = np.zeros(h * w, np.uint8)
        v = int(rng.integers(0, 16))
        for i in range(h * w):
            if rng.random() < p:
                v = int(rng.integers(0, 16))
            img[i] = v
        img = img.reshape(h, w)
        pk = O.orc_nic_compress(img)
        assert pk == O.ref_nic_compress(img), k
        assert np.array_equal(O.ref_nic_decompress(pk, w, h), img)
    # beyond the reference's uint16 domain its decoder still reads our split tokens
    blank = np.full((240, 320), 6, np.uint8)
    assert np.array_equal(O.ref_nic_decompress(O.orc_nic_compress(blank), 320, 240), blank)


@pytest.mark.skipif(O.ref_fdf_lib() is None, reason="libref_fdf.so not built (needs AVX-512VL/BW)")
def test_fdf_random_streams_equal_reference():
    from remap_amd.synth import with_sprites

    for seed, (n, w, h, dens) in enumerate([(14, 128, 64, 25), (10, 160, 96, 80), (12, 224, 64, 12), (8, 320, 128, 150)]):
        fr, _ = scroller_v1(n, width=w, height=h, dens=dens)
        fr = with_sprites(fr, n_sprites=2 + seed % 3, size=7 + 2 * seed, seed=40 + seed)
        r, o = O.ref_fdf(fr), O.orc_fdf_filter(fr)
        assert r["nfrag"] == o["nfrag"] and np.array_equal(r["frag"], o["frag"])
        assert np.array_equal(r["masks"] != 0, o["masks"] != 0), seed
        assert np.array_equal(r["ncontours"], o["ncontours"]), seed
        assert np.array_equal(r["dots"], o["canvases"][0]), seed
