"""The CPU oracle against the golden vectors the REFERENCE produced (oracle/make_golden.py).
Runs without a GPU and without /root/reference."""
import hashlib
import json
import os

import numpy as np
import pytest

import orclib as O
from remap_amd.synth import rooms_v1, scroller_v1

G = os.path.join(os.path.dirname(__file__), "golden")


def desc_of(a):
    return np.ascontiguousarray(a).reshape(-1).view(O.DESC_DTYPE)


def poly_sum(canvas):
    """The checksum ref_harness_full.cpp computes: s = s * 1000003 + v over all bins, mod 2^64."""
    v = np.ascontiguousarray(canvas).reshape(-1).astype(np.uint64)
    p = np.full(len(v), 1000003, np.uint64)
    p[0] = 1
    pw = np.cumprod(p)[::-1]  # p^(N-1-i), wraps modulo 2^64
    with np.errstate(over="ignore"):
        return int((v * pw).sum(dtype=np.uint64))


def test_luts_and_hashes():
    g = json.load(open(os.path.join(G, "luts_hashes.json")))
    n2o, o2n = O.orc_luts()
    assert n2o.tolist() == g["native_to_ordered"] == [0, 15, 2, 12, 6, 9, 3, 13, 5, 1, 7, 4, 8, 14, 10, 11]
    assert o2n.tolist() == g["ordered_to_native"] == [0, 9, 2, 6, 11, 8, 4, 10, 12, 5, 14, 15, 3, 7, 13, 1]
    for c, h in zip(g["hash_codes"], g["hash_code_values"]):
        assert int(O.oracle().orc_hash_code(O.ptr(np.array(c, np.uint8)))) == h
    for (x, y), h in zip(g["hash_offsets"], g["hash_offset_values"]):
        assert int(O.oracle().orc_hash_offset(x, y)) == h


def test_kpe_cases():
    g = np.load(os.path.join(G, "kpe_cases.npz"))
    n = 0
    for i in range(10):
        if f"img{i}" not in g:
            continue
        img = g[f"img{i}"]
        for single, (gw, gh, ov) in ((0, (4, 2, 16)), (1, (1, 1, 0))):
            if f"med{i}_{single}" not in g:
                continue
            med, desc, k = O.orc_kpe(img, gw, gh, ov)
            assert np.array_equal(med, g[f"med{i}_{single}"]), (i, single)
            assert desc.view(np.uint8).reshape(-1, 18).tobytes() == g[f"desc{i}_{single}"].tobytes(), (i, single)
            n += 1
    assert n >= 10


def test_kpe_survey_facts():
    # processed bounds, band limits, the verified code bytes
    img = np.full((240, 320), 6, np.uint8)
    patch = np.array([[6, 6, 6, 6, 6], [6, 2, 2, 2, 6], [6, 2, 5, 2, 6], [6, 2, 2, 2, 6], [6, 6, 6, 6, 6]], np.uint8)
    img[98:103, 80:85] = patch
    med, desc, k = O.orc_kpe(img)
    centre = desc[(desc["x"] == 82) & (desc["y"] == 100)]
    assert len(centre) == 1
    assert bytes(centre[0]["code"]).hex() == "66666622622625662262666662"
    assert centre[0]["region_mask"] == (1 << 0) | (1 << 2)  # x 82 in c0 [2,90) and c1 [74,178); y 100 in r0 only
    rnd = np.random.default_rng(0).integers(0, 16, size=(240, 320), dtype=np.uint8)
    med, desc, k = O.orc_kpe(rnd)
    assert desc["x"].min() == 2 and desc["x"].max() == 317 and desc["y"].min() == 2 and desc["y"].max() == 235
    assert (med[:2] == 0).all() and (med[236:] == 0).all() and (med[:, :2] == 0).all() and (med[:, 318:] == 0).all()
    # band limits verified by the survey for 320x240, grid 4x2, overlap 16
    xb = [(2, 90), (74, 178), (162, 266), (250, 318)]
    yb = [(2, 131), (115, 236)]
    want = np.zeros(len(desc), np.uint8)
    for c, (x0, x1) in enumerate(xb):
        for r, (y0, y1) in enumerate(yb):
            inside = (desc["x"] >= x0) & (desc["x"] < x1) & (desc["y"] >= y0) & (desc["y"] < y1)
            want |= (inside.astype(np.uint8) << (2 * c + r))
    assert np.array_equal(desc["region_mask"], want)
    assert {1, 2, 3, 5, 10, 15, 64, 128, 192}.issubset(set(want.tolist()))  # 1-, 2- and 4-region points occur


def test_kpm_grid_cases():
    g = np.load(os.path.join(G, "kpm_grid_cases.npz"))
    ties = 0
    for i in range(int(g["n"])):
        m, d, t = O.orc_kpm_grid(desc_of(g[f"prev{i}"]), desc_of(g[f"cur{i}"]))
        assert [m, d[0], d[1]] == g[f"res{i}"].tolist(), i
        assert t.view(np.uint32).reshape(8, 3, 4).tobytes() == g[f"tick{i}"].tobytes(), i
        for r in range(8):
            c = [int(v["count"]) for v in t[r] if v["valid"]]
            ties += any(c[j] == c[j + 1] for j in range(len(c) - 1))
    assert ties >= 10  # the fixtures pin libstdc++'s iteration order on real ties


def test_kpm_cell_cases():
    g = np.load(os.path.join(G, "kpm_cell_cases.npz"))
    hit = 0
    for k in range(int(g["npairs"])):
        m, dx, dy, cnt, i, j = g[f"res{k}"].tolist()
        om, ov = O.orc_kpm_cell(desc_of(g[f"desc{i}"]), g[f"mask{i}"], desc_of(g[f"desc{j}"]), g[f"mask{j}"].shape)
        assert om == m and (not m or (ov["dx"], ov["dy"], ov["count"]) == (dx, dy, cnt)), k
        hit += m
    assert hit >= 2


def test_fgm_cases():
    g = np.load(os.path.join(G, "fgm_cases.npz"))
    fr = g["frames"]
    for case in (0, 1):
        pos = g[f"pos{case}"]
        for i in range(1, len(pos) + 1):
            assert O.orc_fgm_extent(pos[:i], 64, 32) == tuple(g[f"trace{case}"][i - 1].tolist()), (case, i)
        cw, ch, zx, zy = O.orc_fgm_extent(pos, 64, 32)
        can = np.zeros((ch, cw, 16), np.uint16)
        for i, p in enumerate(pos):
            O.orc_fgm_blit(can, (zx, zy), fr[i], p)
        assert np.array_equal(can, g[f"dots{case}"])
        img, mask = O.orc_fgm_blend(can)
        assert np.array_equal(img, g[f"blend_img{case}"]) and np.array_equal(mask, g[f"blend_mask{case}"])
    # merge: dst.blit(dst.zero + at, src), then normalize
    a, b = g["merge_a_dots"].copy(), g["merge_b_dots"]
    aw, ah, azx, azy = g["merge_a_info"].tolist()
    at = g["merge_at"]
    # ensure(): the destination grows like a blit of a frame of src's size at `pos` with step = frame dims
    step = (64, 32)
    px, py = azx + int(at[0]), azy + int(at[1])
    sw, sh = b.shape[1], b.shape[0]
    w, h, zx, zy = aw, ah, azx, azy

    def grow(change, s):
        return (change + s - 1) // s * s
    l = grow(zx - px, step[0]) if px < zx else 0
    r = grow(px + sw - (zx + w), step[0]) if px + sw > zx + w else 0
    t = grow(zy - py, step[1]) if py < zy else 0
    bb = grow(py + sh - (zy + h), step[1]) if py + sh > zy + h else 0
    dst = np.zeros((h + t + bb, w + l + r, 16), np.uint16)
    dst[t:t + h, l:l + w] = a
    zx, zy = zx - l, zy - t
    O.orc_fgm_merge(dst, b, (px - zx, py - zy))
    assert np.array_equal(dst, g["merge_dots"])
    assert [dst.shape[1], dst.shape[0]] == g["merge_info"][:2].tolist()


@pytest.mark.parametrize("name", ["config1", "dense"])
def test_frc_streams(name):
    g = np.load(os.path.join(G, "frc_streams.npz"))
    n, dens = g[f"{name}_n"].tolist()
    fr, truth = scroller_v1(n, dens=dens)
    if name == "dense":
        fr = fr.copy()
        fr[20] = 6
        fr[41] = 6
    assert hashlib.sha256(fr.tobytes()).digest() == g[f"{name}_input_sha256"].tobytes(), "generator drifted"
    o = O.orc_frc(fr)
    assert np.array_equal(o["frag"], g[f"{name}_frag"])
    nfrag = int(o["frag"].max()) + 1
    assert nfrag == len(g[f"{name}_dims"])
    for f in range(nfrag):
        sel = o["frag"] == f
        cw, ch, zx, zy = O.orc_fgm_extent(o["pos"][sel], 320, 240)
        assert [cw, ch] == g[f"{name}_dims"][f].tolist()
        assert np.array_equal(o["pos"][sel] - np.array([zx, zy]), g[f"{name}_pos_norm"][sel])
        can = np.zeros((ch, cw, 16), np.uint16)
        for i in np.nonzero(sel)[0]:
            O.orc_fgm_blit(can, (zx, zy), fr[i], o["pos"][i])
        assert poly_sum(can) == int(g[f"{name}_canvas_sum"][f])
    if name == "config1":  # BASELINE.json configs[0]: 99 offsets, 1 fragment, all equal to the camera truth
        assert nfrag == 1 and o["matched"][1:].all() and np.array_equal(o["dxdy"][1:], truth[1:])


def test_fgs_input_is_reproducible():
    g = np.load(os.path.join(G, "fgs_splice.npz"))
    n, w, h, rooms, dens, seed = g["gen"].tolist()
    fr, _, _ = rooms_v1(n, width=w, height=h, n_rooms=rooms, dens=dens, seed=seed)
    assert hashlib.sha256(fr.tobytes()).digest() == g["input_sha256"].tobytes()
    o = O.orc_frc(fr)
    assert int(o["frag"].max()) + 1 == int(g["ncollected"])


def test_rooms_stream_collect_and_votes():
    """tests/golden/fgs_splice_rooms.npz (configs[4] in small, written by the reference: 80 fragments, 16
    merges): the oracle's collect gives the reference's fragments, and its cellular match the reference's
    votes — all 16 valid pairs and a sample of the 3 144 others."""
    from remap_amd.synth import rooms_v1

    g = np.load(os.path.join(G, "fgs_splice_rooms.npz"))
    n, w, h, rooms, dens, seed = g["gen"].tolist()
    fr, _, _ = rooms_v1(n, width=w, height=h, n_rooms=rooms, dens=dens, seed=seed)
    assert hashlib.sha256(np.ascontiguousarray(fr).tobytes()).digest() == g["input_sha256"].tobytes()
    o = O.orc_frc(fr)
    assert np.array_equal(o["frag"], g["collected_frag"]) and int(g["ncollected"]) == 80 and int(g["nfrag"]) == 64
    nf = int(g["ncollected"])
    pairs = [(i, j) for i in range(nf) for j in range(i + 1, nf)]
    valid = np.nonzero(g["votes"][:, 3])[0]
    assert len(valid) == 16
    rng = np.random.default_rng(5)
    sample = sorted(set(valid.tolist()) | set(rng.choice(len(pairs), 60, replace=False).tolist()))
    tabs = {}

    def table(f):
        if f not in tabs:
            idx = np.nonzero(o["frag"] == f)[0]
            cw, ch, zx, zy = O.orc_fgm_extent(o["pos"][idx], w, h)
            assert [cw, ch] == g["collected_dims"][f].tolist()
            can = np.zeros((ch, cw, 16), np.uint16)
            for i in idx:
                O.orc_fgm_blit(can, (zx, zy), fr[i], o["pos"][i])
            img, mask = O.orc_fgm_blend(can)
            desc = O.orc_kpe(img, 1, 1, 0)[1]
            assert len(desc) == int(g["kcounts"][f])
            tabs[f] = (mask, desc)
        return tabs[f]

    for k in sample:
        i, j = pairs[k]
        m, v = O.orc_kpm_cell(table(i)[1], table(i)[0], table(j)[1], table(j)[0].shape)
        want = g["votes"][k]
        assert int(m) == int(want[3]), (i, j)
        if m:
            assert (int(v["dx"]), int(v["dy"]), int(v["count"])) == tuple(int(x) for x in want[:3]), (i, j)


def test_ful_fragment_files():
    """remap_amd.ful pack/unpack against files the reference's ful::write produced."""
    from remap_amd import ful

    g = np.load(os.path.join(G, "ful_case.npz"))
    for k in range(int(g["nfiles"])):
        buf = g[f"file{k}"].tobytes()
        dots, zero, frames = ful.unpack(buf)
        w, h, zx, zy, nf = g[f"info{k}"].tolist()
        assert dots.shape == (h, w, 16) and np.array_equal(dots, g[f"dots{k}"])
        assert zero == (zx, zy) and len(frames) == nf
        sizes, blob = g[f"blob_sizes{k}"], g[f"blob_bytes{k}"].tobytes()
        o = 0
        for i, (no, pos, bi, bm) in enumerate(frames):
            assert no == int(g[f"numbers{k}"][i]) and list(pos) == g[f"pos{k}"][i].tolist()
            ni, nm = sizes[i].tolist()
            assert bi == blob[o:o + ni] and bm == blob[o + ni:o + ni + nm]
            o += ni + nm
        assert ful.pack(dots, zero, frames) == buf  # byte-identical to the reference's file


def nic_cases():
    g = np.load(os.path.join(G, "nic_cases.npz"))
    po, ko = 0, 0
    for k in range(int(g["ncases"])):
        h, w = g["shapes"][k].tolist()
        img = g["pixels"][po:po + h * w].reshape(h, w)
        pk = g["packed"][ko:ko + int(g["sizes"][k])].tobytes()
        po += h * w
        ko += int(g["sizes"][k])
        yield k, img, pk


def test_nic_compress_cases():
    """oracle nic codec == the reference's nic::compress bytes on every golden case; decode inverts."""
    n = 0
    for k, img, pk in nic_cases():
        assert O.orc_nic_compress(img) == pk, k
        d, cnt = O.orc_nic_decompress(pk, img.size)
        assert cnt == img.size and np.array_equal(d, img.reshape(-1)), k
        n += 1
    assert n > 200


def test_nic_beyond_the_reference_domain_round_trips():
    """runs > 65535 pixels and literals > 16383 (where the reference overflows its uint16
    counters) are split into well-formed tokens that decode to the input."""
    rng = np.random.default_rng(8)
    blank = np.full(76800, 6, np.uint8)
    noise = rng.integers(0, 16, size=76800, dtype=np.uint8)
    for r in (65536, 65537, 65538, 131070, 131071, 131073):
        x = np.concatenate([[1, 2], np.full(r, 9, np.uint8), [3]]).astype(np.uint8)
        pk = O.orc_nic_compress(x)
        d, cnt = O.orc_nic_decompress(pk, x.size)
        assert cnt == x.size and np.array_equal(d, x), r
    for x in (blank, noise):
        pk = O.orc_nic_compress(x)
        d, cnt = O.orc_nic_decompress(pk, x.size)
        assert cnt == x.size and np.array_equal(d, x)
    assert O.orc_nic_compress(blank).hex() == "66ffff66012c"  # 65535 + 11265


def fdf_streams():
    """The streams of tests/golden/fdf_cases.npz, regenerated from the committed generator."""
    from remap_amd.synth import rooms_v1, with_sprites

    a, _ = scroller_v1(24, width=128, height=64, dens=30)
    a = with_sprites(a, n_sprites=3, size=11, seed=5)
    a[10] = 6
    b, _, _ = rooms_v1(60, width=128, height=96, n_rooms=2, dens=40, seed=3)
    b = with_sprites(b, n_sprites=2, size=9, seed=9)
    c, _ = scroller_v1(8, width=320, height=240, dens=200)
    c = with_sprites(c, n_sprites=3, size=15, seed=1)
    return [a, b, c]


def test_fdf_cases():
    """oracle collect + foreground filter == the reference's fdf::filter: per-frame masks,
    kept contours, filtered canvases (checksums)."""
    g = np.load(os.path.join(G, "fdf_cases.npz"))
    for k, fr in enumerate(fdf_streams()):
        assert hashlib.sha256(fr.tobytes()).digest() == g[f"input_sha256_{k}"].tobytes()
        o = O.orc_fdf_filter(fr)
        n, H, W = fr.shape
        want = np.unpackbits(g[f"masks{k}"])[: n * H * W].reshape(n, H, W)
        assert np.array_equal(o["masks"] != 0, want != 0), k
        assert np.array_equal(o["ncontours"], g[f"ncontours{k}"]) and np.array_equal(o["frag"], g[f"frag{k}"]), k
        assert o["nfrag"] == len(g[f"dims{k}"])
        for f, can in enumerate(o["canvases"]):
            assert [can.shape[1], can.shape[0]] == g[f"dims{k}"][f].tolist()
            assert poly_sum(can) == int(g[f"canvas_sum{k}"][f]), (k, f)
