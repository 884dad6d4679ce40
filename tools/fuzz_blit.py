"""Random shapes through the three blit variants (fgm_blit with its lists in residue order / as binned, the bit-plane
kernel): the canvases must be equal bit for bit, fresh and adding, and equal to the oracle's for the small cases.
usage: fuzz_blit.py [cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
from remap_amd import lib as L

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 24
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for case in range(cases):
    W = int(rng.integers(64, 420)); H = int(rng.integers(5, 270))
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
    if cw * ch * 32 > 3e9:
        ctx.close(); continue
    got = {}
    for name, bs, st in (("lds", "0", "32"), ("lds-unsorted", "0", "0"), ("bitplanes", "1", "32")):
        os.environ["RMX_BLIT_BS"], os.environ["RMX_BLIT_STRIDE"] = bs, st
        for fresh in (False, True):
            canvas = torch.full((ch, cw, 16), 0 if not fresh else 0x3333, dtype=torch.int16, device="cuda")
            ctx.fgm_blit(d, dp, canvas, (zx, zy), fresh=fresh)
            if not fresh:
                ctx.fgm_blit(d, dp, canvas, (zx, zy))  # (a second call adds)
            got[(name, fresh)] = canvas
    ref = got[("lds", False)]
    ok = all(torch.equal(got[(nm, False)], ref) for nm in ("lds-unsorted", "bitplanes"))
    half = got[("lds", True)]
    ok = ok and all(torch.equal(got[(nm, True)], half) for nm in ("lds-unsorted", "bitplanes"))
    ok = ok and torch.equal((half.to(torch.int32) * 2).to(torch.int16), ref)
    if n <= 40:
        import orclib as O
        want = np.zeros((ch, cw, 16), np.uint16)
        for i in range(n):
            O.orc_fgm_blit(want, (zx, zy), frames[idx[i]], pos[i])
        ok = ok and np.array_equal(half.cpu().numpy().view(np.uint16), want)
    print(f"case {case}: {W}x{H} n={n} step={step} canvas {cw}x{ch}: {'ok' if ok else 'MISMATCH'}", flush=True)
    bad += not ok
    ctx.close()
print("mismatches:", bad)
sys.exit(1 if bad else 0)
