"""fdf::filter on the GPU (rmx_fdf_filter_batch through remap_amd.fdf) against the reference's
golden masks / contour counts / filtered canvases and against the CPU oracle."""
import os

import numpy as np
import pytest

import orclib as O
from remap_amd import lib as L
from remap_amd.fdf import filter_fragments
from remap_amd.pipeline import Aligner
from remap_amd.synth import scroller_v1, with_sprites
from test_oracle_golden import fdf_streams, poly_sum

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def gpu_filter(fr):
    import torch

    n, H, W = fr.shape
    al = Aligner(W, H, kmax=2048)
    d = torch.from_numpy(fr).cuda()
    out = al.align(d, blit=True)
    frags = al.fragments(d, out)
    res, info = filter_fragments(frags, d, out["median"], W, H, ctx=al.ctx, want_masks=True)
    masks = np.stack([info[i][0].cpu().numpy() for i in range(n)])
    ncont = np.array([info[i][1] for i in range(n)])
    frag = out["frag"].cpu().numpy()
    al.close()
    return res, masks, ncont, frag


def test_fdf_golden_streams(gpu):
    g = np.load(os.path.join(G, "fdf_cases.npz"))
    for k, fr in enumerate(fdf_streams()):
        res, masks, ncont, frag = gpu_filter(fr)
        n, H, W = fr.shape
        want = np.unpackbits(g[f"masks{k}"])[: n * H * W].reshape(n, H, W)
        assert np.array_equal(masks != 0, want != 0), k
        assert np.array_equal(ncont, g[f"ncontours{k}"]) and np.array_equal(frag, g[f"frag{k}"]), k
        assert len(res) == len(g[f"dims{k}"])
        for f, fgm in enumerate(res):
            assert list(fgm.dims) == g[f"dims{k}"][f].tolist()
            assert poly_sum(fgm.canvas.cpu().numpy().view(np.uint16)) == int(g[f"canvas_sum{k}"][f]), (k, f)


def test_fdf_matches_oracle_on_other_sizes(gpu):
    # widths the reference's aligned loads cannot take (not multiples of 32), big flat
    # components over the area limit, frames without any foreground
    for (n, w, h, dens, sprites) in ((12, 200, 100, 15, 3), (10, 322, 130, 400, 0), (6, 320, 240, 40, 4)):
        fr, _ = scroller_v1(n, width=w, height=h, dens=dens)
        if sprites:
            fr = with_sprites(fr, n_sprites=sprites, size=13, seed=n)
        res, masks, ncont, frag = gpu_filter(fr)
        o = O.orc_fdf_filter(fr)
        assert np.array_equal(masks != 0, o["masks"] != 0), (w, h)
        assert np.array_equal(ncont, o["ncontours"]) and np.array_equal(frag, o["frag"])
        for f, can in enumerate(o["canvases"]):
            assert np.array_equal(res[f].canvas.cpu().numpy().view(np.uint16), can), (w, h, f)
