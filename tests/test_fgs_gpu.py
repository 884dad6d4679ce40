"""fgs: cellular kpm::match and the whole collect -> splice chain on the GPU, against the
oracle and the vectors the reference produced (tests/golden/kpm_cell_cases.npz, fgs_splice.npz)."""
import os

import numpy as np
import pytest

import orclib as O
from remap_amd import lib as L
from remap_amd.pipeline import Aligner
from remap_amd.splice import Fragment, splice
from remap_amd.synth import rooms_v1
from test_oracle_golden import desc_of, poly_sum

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("big", [False, True])
def test_cellular_match_golden_pairs(gpu, big, monkeypatch):
    """big: every pair through the global-memory instance (fgs_big.hip) instead of the LDS one"""
    import torch

    if big:
        monkeypatch.setenv("RMX_FGS_BIG", "1")

    g = np.load(os.path.join(G, "kpm_cell_cases.npz"))
    nf = int(g["nfrag"])
    ctx = L.Context(16, 8, grid=(1, 1), overlap=0, kmax=2048)
    tabs, ks, masks, dims = [], [], [], []
    for i in range(nf):
        d = g[f"desc{i}"]
        t = torch.zeros((2048, 18), dtype=torch.uint8, device="cuda")
        t[: len(d)] = torch.from_numpy(d).cuda()
        tabs.append(t)
        ks.append(len(d))
        m = g[f"mask{i}"]
        masks.append(torch.from_numpy(m).cuda())
        dims.append((m.shape[1], m.shape[0]))
    pairs, want = [], []
    for k in range(int(g["npairs"])):
        m, dx, dy, cnt, i, j = g[f"res{k}"].tolist()
        pairs.append((i, j))
        want.append((m, dx, dy, cnt))
    votes = ctx.kpm_cell_pairs(tabs, ks, masks, dims, pairs)
    for v, (m, dx, dy, cnt), (i, j) in zip(votes, want, pairs):
        assert int(v["valid"]) == m, (i, j)
        if m:
            assert (int(v["dx"]), int(v["dy"]), int(v["count"])) == (dx, dy, cnt)
        # and the oracle agrees (same inputs)
        om, ov = O.orc_kpm_cell(desc_of(g[f"desc{i}"]), g[f"mask{i}"], desc_of(g[f"desc{j}"]), g[f"mask{j}"].shape)
        assert om == m
    ctx.close()


@pytest.mark.parametrize("big", [False, True])
def test_collect_then_splice_matches_reference(gpu, big, monkeypatch):
    import torch

    if big:
        monkeypatch.setenv("RMX_FGS_BIG", "1")

    g = np.load(os.path.join(G, "fgs_splice.npz"))
    n, w, h, rooms, dens, seed = g["gen"].tolist()
    frames, _, _ = rooms_v1(n, width=w, height=h, n_rooms=rooms, dens=dens, seed=seed)
    al = Aligner(w, h, kmax=2048)
    out = al.align(torch.from_numpy(frames).cuda(), blit=True)
    pos = out["pos"].cpu().numpy()
    frag = out["frag"].cpu().numpy()
    assert out["nfrag"] == int(g["ncollected"])
    frags = []
    for fid in range(out["nfrag"]):
        canvas, zero, _ = out["canvases"][fid]
        idx = np.nonzero(frag == fid)[0]
        f = Fragment(canvas.clone(), zero, (w, h), [(int(i), pos[i]) for i in idx])
        f.normalize()  # collector.complete(), frc.hpp:74-80
        frags.append(f)
    res = splice(frags)
    assert len(res) == int(g["nfrag"])
    for k, f in enumerate(res):
        assert list(f.dims) == g["dims"][k].tolist(), k
        assert poly_sum(f.canvas.cpu().numpy().view(np.uint16)) == int(g["canvas_sum"][k]), k
        for no, p in f.frames:
            assert int(g["frag"][no]) == k and list(p) == g["pos"][no].tolist(), (k, no)
    al.close()


def test_rooms_stream_votes_and_splice_equal_the_reference(gpu):
    """BASELINE.json configs[4] in small (tests/golden/fgs_splice_rooms.npz, written by the compiled
    reference): 2 000 frames of rooms-v1 at 320x240 -> 80 fragments -> 16 merges.  Collect on the GPU gives
    the reference's fragments; the batched extract_all + rmx_kpm_cell_pairs give the reference's vote for
    every one of the 3 160 initial pairs (kpm.hpp:371-393); fgs::splice (fgs.hpp:187-213: sixteen rounds
    of select_match / splice_single / match_partial) ends in the reference's canvases and positions."""
    import torch

    from remap_amd import splice as S

    g = np.load(os.path.join(G, "fgs_splice_rooms.npz"))
    n, w, h, rooms, dens, seed = g["gen"].tolist()
    frames, _, _ = rooms_v1(n, width=w, height=h, n_rooms=rooms, dens=dens, seed=seed)
    al = Aligner(w, h, kmax=2048)
    d = torch.from_numpy(frames).cuda()
    out = al.align(d, blit=True)
    assert out["nfrag"] == int(g["ncollected"]) and np.array_equal(out["frag"].cpu().numpy(), g["collected_frag"])
    frags = al.fragments(d, out)
    assert [list(f.dims) for f in frags] == g["collected_dims"].tolist()
    # the vote matrix of match_all
    ctxs = {}

    def ctx_cache(key):
        if key not in ctxs:
            cw, ch, kmax = (16, 8, S.CELL_CTX_KMAX) if key == "any" else key
            ctxs[key] = L.Context(cw, ch, grid=(1, 1), overlap=0, kmax=kmax)
        return ctxs[key]

    snips = S._extract_all(ctx_cache, frags)
    assert [s.k for s in snips] == g["kcounts"].tolist()
    pairs = [(i, j) for i in range(len(snips)) for j in range(i + 1, len(snips))]
    votes = ctx_cache("any").kpm_cell_pairs([s.table for s in snips], [s.k for s in snips], [s.mask for s in snips],
                                            [s.fragment.dims for s in snips], pairs, S.CELL)
    got = np.stack([votes["dx"] * votes["valid"], votes["dy"] * votes["valid"], votes["count"] * votes["valid"], votes["valid"]], axis=1)
    assert np.array_equal(got.astype(np.int64), g["votes"].astype(np.int64))
    for c in ctxs.values():
        c.close()
    res = splice(frags)
    assert len(res) == int(g["nfrag"]) and int(g["ncollected"]) - len(res) == 16
    for k, f in enumerate(res):
        assert list(f.dims) == g["dims"][k].tolist(), k
        assert poly_sum(f.canvas.cpu().numpy().view(np.uint16)) == int(g["canvas_sum"][k]), k
        for no, p in f.frames:
            assert int(g["frag"][no]) == k and list(p) == g["pos"][no].tolist(), (k, no)
    al.close()


def _collect_one(frames, w, h):
    """one fragment out of a frame run: device Fragment + the oracle's canvas and positions"""
    import torch

    al = Aligner(w, h, kmax=4096)
    out = al.align(torch.from_numpy(frames).cuda(), blit=True)
    assert out["nfrag"] == 1
    canvas, zero, _ = out["canvases"][0]
    pos = out["pos"].cpu().numpy()
    f = Fragment(canvas.clone(), zero, (w, h), [(i, pos[i]) for i in range(len(frames))])
    f.normalize()
    al.close()
    o = O.orc_frc(frames)
    cw, ch, zx, zy = O.orc_fgm_extent(o["pos"], w, h)
    can = np.zeros((ch, cw, 16), np.uint16)
    for i in range(len(frames)):
        O.orc_fgm_blit(can, (zx, zy), frames[i], o["pos"][i])
    assert np.array_equal(f.canvas.cpu().numpy().view(np.uint16), can)
    return f, can, o["pos"] - [zx, zy]


def test_splice_fragments_with_more_than_2048_keypoints(gpu):
    """Two overlapping runs of a dense scroller: each fragment's blended canvas has between 2 048
    and 4 096 keypoints, more than the snippet tables start with — extract_single repeats the
    extraction with larger tables (the reference has no limit, kpr.hpp:121-124).  Vote, merged
    canvas and frame positions equal what the oracle's pieces give."""
    from remap_amd.synth import scroller_v1

    w, h = 320, 240
    frames, _ = scroller_v1(300, dens=30)
    parts = [frames[:170], frames[130:]]  # 3 519 and 3 317 keypoints, 4 750 once merged
    frags, cans, poss = zip(*[_collect_one(p, w, h) for p in parts])
    descs, masks = [], []
    for can in cans:
        img, mask = O.orc_fgm_blend(can)
        _, d, k = O.orc_kpe(img, 1, 1, 0)
        assert k > 2048, k
        descs.append(d)
        masks.append(mask)
    m, v = O.orc_kpm_cell(descs[0], masks[0], descs[1], masks[1].shape)
    assert m == 1
    off = (int(v["dx"]), int(v["dy"]))
    res = splice(list(frags))
    assert len(res) == 1
    f = res[0]
    # frames: the first fragment's stay, the second's move by the vote's offset; then normalize
    allpos = np.array([p for _, p in f.frames])
    exp = np.concatenate([poss[0], poss[1] + off])
    # canvas growth happens in whole frame steps (fgm.hpp:190-233): the merged zero is a step multiple away
    shift = allpos[0] - exp[0]
    assert shift[0] % w == 0 and shift[1] % h == 0
    assert np.array_equal(allpos, exp + shift)
    ch, cw = f.canvas.shape[0], f.canvas.shape[1]
    assert cw % w == 0 and ch % h == 0
    ref = np.zeros((ch, cw, 16), np.uint16)
    allframes = np.concatenate(parts)
    for img, p in zip(allframes, allpos):
        O.orc_fgm_blit(ref, (0, 0), img, p)
    assert np.array_equal(f.canvas.cpu().numpy().view(np.uint16), ref)
    # minimal: no empty frame-sized column or row at either side
    occ = ref.any(axis=2)
    assert occ[:, :w].any() and occ[:, -w:].any() and occ[:h].any() and occ[-h:].any()


def _random_table(rng, k, ncodes, w, h):
    """k keypoints at distinct positions of a w x h canvas, codes drawn from ncodes patterns; rows in
    scan order (x, then y) as kpe emits them"""
    pos = rng.choice(w * h, size=k, replace=False)
    x, y = pos // h, pos % h
    order = np.lexsort((y, x))
    codes = rng.integers(0, 256, size=(ncodes, 13), dtype=np.uint8)
    codes[:, 12] = (codes[:, 12] & 0xF0) | rng.integers(1, 3, size=ncodes, dtype=np.uint8)
    d = np.zeros(k, O.DESC_DTYPE)
    d["code"] = codes[rng.integers(0, ncodes, size=k)]
    d["region_mask"] = 1
    d["x"] = x[order]
    d["y"] = y[order]
    return d


@pytest.mark.parametrize("case", [(6000, 5000, 300, 1), (9000, 7000, 4000, 2), (5000, 300, 40, 3), (20000, 18000, 15000, 4),
                                  (6000, 6000, 200000, 5)])  # (5: a handful of shared codes, a thousand active cells)
def test_cellular_match_large_tables_match_oracle(gpu, case):
    """Fragments above the LDS instance's 4 096 keypoints (fgs_big.hip) against the oracle: few
    codes (many point pairs per code, thousands of distinct offsets, ties at the top resolved in
    libstdc++'s iteration order) and many codes; previous and current swapped as well."""
    import torch

    kp, kc, ncodes, seed = case
    rng = np.random.default_rng(seed)
    dims = [(700, 500), (640, 480)]
    tabs = [_random_table(rng, kp, ncodes, *dims[0]), _random_table(rng, kc, ncodes, *dims[1])]
    masks = [(rng.random((h, w)) < 0.7).astype(np.uint8) for (w, h) in dims]
    ctx = L.Context(16, 8, grid=(1, 1), overlap=0, kmax=4096)
    dt = [torch.from_numpy(t.view(np.uint8).reshape(-1, 18)).cuda() for t in tabs]
    dm = [torch.from_numpy(m).cuda() for m in masks]
    votes = ctx.kpm_cell_pairs(dt, [len(t) for t in tabs], dm, dims, [(0, 1), (1, 0)])
    for v, (i, j) in zip(votes, [(0, 1), (1, 0)]):
        m, ov = O.orc_kpm_cell(tabs[i], masks[i], tabs[j], masks[j].shape)
        assert int(v["valid"]) == m, (case, i, j)
        if m:
            assert (int(v["dx"]), int(v["dy"]), int(v["count"])) == (int(ov["dx"]), int(ov["dy"]), int(ov["count"])), (case, i, j)
    ctx.close()


def test_cellular_match_shifted_copy_large(gpu):
    """a real overlap: the current table is the previous one shifted by (37, -12) plus clutter —
    the vote must be that offset with every shared keypoint's count"""
    import torch

    rng = np.random.default_rng(11)
    w, h = 900, 600
    base = _random_table(rng, 7000, 6500, w, h)
    cur = base.copy()
    keep = (cur["x"] >= 37) & (cur["y"] + 12 < h)
    cur = cur[keep]
    cur["x"] -= 37
    cur["y"] += 12
    order = np.lexsort((cur["y"], cur["x"]))
    cur = np.ascontiguousarray(cur[order])
    masks = [np.ones((h, w), np.uint8), np.ones((h, w), np.uint8)]
    ctx = L.Context(16, 8, grid=(1, 1), overlap=0, kmax=4096)
    dt = [torch.from_numpy(t.view(np.uint8).reshape(-1, 18)).cuda() for t in (base, cur)]
    dm = [torch.from_numpy(m).cuda() for m in masks]
    v = ctx.kpm_cell_pairs(dt, [len(base), len(cur)], dm, [(w, h), (w, h)], [(0, 1)])[0]
    m, ov = O.orc_kpm_cell(base, masks[0], cur, masks[1].shape)
    assert m == 1 and int(v["valid"]) == 1
    assert (int(v["dx"]), int(v["dy"])) == (37, -12) == (int(ov["dx"]), int(ov["dy"]))
    assert int(v["count"]) == int(ov["count"])
    ctx.close()
