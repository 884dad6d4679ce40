"""Random shapes through the three blit variants (fgm.hip's kernel with its frame lists in residue order and as the
binning atomics leave them, the bit-plane kernel of fgm_bs.hip): the canvases must agree bit for bit — written fresh
and added twice — and equal the oracle's where that is quick (reference fgm.hpp:87-97)."""
import os

import numpy as np
import pytest

import orclib as O
from remap_amd import lib as L

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("seed", [3, 11])
def test_blit_variants_agree_on_random_shapes(gpu, seed, monkeypatch):
    import torch

    rng = np.random.default_rng(seed)
    for case in range(8):
        W, H = int(rng.integers(64, 420)), int(rng.integers(5, 270))
        n = int(rng.choice([3, 40, 700, 1030, 2500, 6000]))
        step = int(rng.choice([0, 1, 3, 12, 60]))
        frames = rng.integers(0, 16, size=(min(n, 64), H, W), dtype=np.uint8)
        idx = rng.integers(0, frames.shape[0], size=n)
        pos = np.cumsum(rng.integers(-step, step + 1, size=(n, 2)), axis=0).astype(np.int32)
        pos[0] = 0
        d = torch.from_numpy(frames).cuda()[torch.from_numpy(idx).cuda()].contiguous()
        dp = torch.from_numpy(pos).cuda()
        ctx = L.Context(W, H, grid=(1, 1), overlap=0, kmax=64)
        cw, ch, zx, zy = ctx.fgm_extent(pos)
        if cw * ch * 32 > 2e9:
            ctx.close()
            continue
        got = {}
        for name, bs, st in (("lds", "0", "32"), ("lds-unsorted", "0", "0"), ("bitplanes", "1", "32")):
            monkeypatch.setenv("RMX_BLIT_BS", bs)
            monkeypatch.setenv("RMX_BLIT_STRIDE", st)
            for fresh in (False, True):
                canvas = torch.full((ch, cw, 16), 0x3333 if fresh else 0, dtype=torch.int16, device="cuda")
                ctx.fgm_blit(d, dp, canvas, (zx, zy), fresh=fresh)
                if not fresh:
                    ctx.fgm_blit(d, dp, canvas, (zx, zy))  # (a second call adds)
                got[(name, fresh)] = canvas
        twice, once = got[("lds", False)], got[("lds", True)]
        for nm in ("lds-unsorted", "bitplanes"):
            assert torch.equal(got[(nm, False)], twice), (case, W, H, n, step, nm, "added twice")
            assert torch.equal(got[(nm, True)], once), (case, W, H, n, step, nm, "fresh")
        assert torch.equal((once.to(torch.int32) * 2).to(torch.int16), twice), (case, W, H, n, step)
        if n <= 40:
            want = np.zeros((ch, cw, 16), np.uint16)
            for i in range(n):
                O.orc_fgm_blit(want, (zx, zy), frames[idx[i]], pos[i])
            assert np.array_equal(once.cpu().numpy().view(np.uint16), want), (case, W, H, n, step)
        ctx.close()
