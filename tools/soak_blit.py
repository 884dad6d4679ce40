"""Randomised fgm_blit against the oracle: frame sizes on both sides of the 64-pixel switch,
batches on both sides of the binning switch (1 024 frames), random-walk and jitter positions,
canvases that already hold counts.  usage: soak_blit.py [seconds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import orclib as O
from remap_amd import lib as L

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t0 = time.time()
cases = bad = 0
while time.time() - t0 < budget:
    W = int(rng.choice([16, 20, 63, 64, 65, 96, 128, 133, 200, 320, 400]))
    H = int(rng.integers(8, 70))
    n = int(rng.choice([1, 2, 7, 40, 300, 1024, 1025, 1500]))
    if W * H * n > 40_000_000:
        n = max(1, 40_000_000 // (W * H))
    frames = rng.integers(0, 16, size=(n, H, W), dtype=np.uint8)
    if rng.random() < 0.5:
        pos = np.cumsum(rng.integers(-5, 6, size=(n, 2)), axis=0).astype(np.int32)
    else:
        pos = rng.integers(-W, W + 1, size=(n, 2)).astype(np.int32)
    pos[0] = 0
    ctx = L.Context(W, H, grid=(1, 1), overlap=0, kmax=64)
    cw, ch, zx, zy = ctx.fgm_extent(pos)
    start = rng.integers(0, 65536, size=(ch, cw, 16)).astype(np.uint16) if rng.random() < 0.3 else np.zeros((ch, cw, 16), np.uint16)
    ocan = start.copy()
    sel = None
    if rng.random() < 0.3 and n > 1:  # only one fragment id of two
        sel = (rng.integers(0, 2, size=n)).astype(np.int32)
    for i in range(n):
        if sel is None or sel[i] == 1:
            O.orc_fgm_blit(ocan, (zx, zy), frames[i], pos[i])
    canvas = torch.from_numpy(start.view(np.int16).copy()).cuda()
    ctx.fgm_blit(torch.from_numpy(frames).cuda(), torch.from_numpy(pos).cuda(), canvas, (zx, zy),
                 frag_id=None if sel is None else torch.from_numpy(sel).cuda(), fragment=1)
    ok = np.array_equal(canvas.cpu().numpy().view(np.uint16), ocan)
    cases += 1
    if not ok:
        bad += 1
        print("MISMATCH", W, H, n, "sel" if sel is not None else "", flush=True)
    ctx.close()
print(f"{cases} cases, {bad} mismatches in {time.time() - t0:.0f} s", flush=True)
