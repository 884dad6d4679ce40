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


def test_cellular_match_golden_pairs(gpu):
    import torch

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


def test_collect_then_splice_matches_reference(gpu):
    import torch

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
