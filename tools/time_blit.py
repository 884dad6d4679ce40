"""fgm_blit alone on the bench stream (n frames at their true positions): wall time per call
(binning kernels included).  usage: time_blit.py [n]; tools/pmc_blit.sh profiles this script."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from remap_amd import lib as L
from remap_amd.synth import scroller_v1

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
shapes = ["default"]
frames, truth = scroller_v1(n, dens=200)
pos = np.cumsum(truth, axis=0).astype(np.int32)
d = torch.from_numpy(frames).cuda()
dp = torch.from_numpy(pos).cuda()
ref = None
for sh in shapes:
    ctx = L.Context(320, 240)
    cw, ch, zx, zy = ctx.fgm_extent(pos)
    canvas = torch.zeros((ch, cw, 16), dtype=torch.int16, device="cuda")
    for _ in range(2):
        ctx.fgm_blit(d, dp, canvas, (zx, zy))
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10):
        ctx.fgm_blit(d, dp, canvas, (zx, zy))
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / 10 * 1e3
    canvas.zero_()
    ctx.fgm_blit(d, dp, canvas, (zx, zy))
    got = canvas.cpu()
    if ref is None:
        ref = got
    print(f"{sh}: {ms:.3f} ms per call  same_as_first={bool(torch.equal(got, ref))}  sum={int(got.view(torch.uint8).sum())}", flush=True)

# the canvas consumers beside it: fgm_blend (32 B read + 2 B written per dot) and fgm_merge
cw, ch, zx, zy = ctx.fgm_extent(pos)
for _ in range(2):
    img, mask = ctx.fgm_blend(canvas)
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(10):
    img, mask = ctx.fgm_blend(canvas)
torch.cuda.synchronize(); ms = (time.perf_counter() - t) / 10 * 1e3
print(f"fgm_blend: {ms:.3f} ms per call, {ch * cw * 34 / ms / 1e6:.0f} GB/s ({ch}x{cw} dots, 34 B each)", flush=True)
big = torch.zeros((ch, cw, 16), dtype=torch.int16, device="cuda")
for _ in range(2):
    ctx.fgm_merge(big, canvas, (0, 0))
torch.cuda.synchronize(); t = time.perf_counter()
for _ in range(10):
    ctx.fgm_merge(big, canvas, (0, 0))
torch.cuda.synchronize(); ms = (time.perf_counter() - t) / 10 * 1e3
print(f"fgm_merge: {ms:.3f} ms per call, {ch * cw * 96 / ms / 1e6:.0f} GB/s ({ch}x{cw} dots, 32 B read + 32 B read + 32 B written)", flush=True)
