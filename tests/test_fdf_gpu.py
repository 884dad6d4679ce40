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


def test_fdf_strip_shapes_match_oracle(gpu):
    """rmx_fdf_filter_batch straight against orc_fdf_mask on shapes that stress fdf_local's strips:
    one strip, interior rows an exact multiple of the strip, two-row and one-row strips (wide
    frames), components that span every strip, the smallest frames a context takes, widths that are no multiple of 4."""
    import torch

    rng = np.random.default_rng(5)
    shapes = [(16, 8), (17, 9), (320, 27), (320, 51), (1280, 20), (3001, 9), (4100, 8), (7680, 8), (641, 37)]
    for (W, H) in shapes:
        n = 3
        # medians: a few flat colours in blocks (big components across strips) plus speckle
        med = np.repeat(np.repeat(rng.integers(0, 3, size=(n, (H + 5) // 6, (W + 8) // 9), dtype=np.uint8), 6, axis=1), 9, axis=2)[:, :H, :W]
        speck = rng.random((n, H, W)) < 0.08
        med = np.where(speck, rng.integers(0, 16, size=(n, H, W), dtype=np.uint8), med).astype(np.uint8)
        med[1, :, W // 2:] = 7  # a component over the area limit
        bw, bh = W + 6, H + 4
        bg = rng.integers(0, 2, size=(bh, bw), dtype=np.uint8)
        frames = np.empty((n, H, W), np.uint8)
        pos = np.array([[0, 0], [6, 4], [3, 1]], np.int32)
        for f in range(n):
            x0, y0 = pos[f]
            frames[f] = bg[y0:y0 + H, x0:x0 + W]
            differ = rng.random((H, W)) < 0.02  # seeds
            frames[f] = np.where(differ, frames[f] ^ 1, frames[f])
        ctx = L.Context(W, H, kmax=64, grid=(1, 1), overlap=0)
        masks, ncont = ctx.fdf_filter(torch.from_numpy(bg).cuda(), torch.from_numpy(frames).cuda(),
                                      torch.from_numpy(med).cuda(), torch.from_numpy(pos).cuda())
        masks, ncont = masks.cpu().numpy(), ncont.cpu().numpy()
        for f in range(n):
            want, k = O.orc_fdf_mask(bg, pos[f], frames[f], med[f])
            assert k == ncont[f], (W, H, f, k, ncont[f])
            assert np.array_equal(masks[f] != 0, want != 0), (W, H, f)
        ctx.close()
    # wider than a strip holds: refused, not mislabelled
    ctx = L.Context(7684, 8, kmax=64, grid=(1, 1), overlap=0)
    z = torch.zeros((1, 8, 7684), dtype=torch.uint8, device="cuda")
    with pytest.raises(L.RemapError):
        ctx.fdf_filter(torch.zeros((8, 7684), dtype=torch.uint8, device="cuda"), z, z, torch.zeros((1, 2), dtype=torch.int32, device="cuda"))
    ctx.close()
