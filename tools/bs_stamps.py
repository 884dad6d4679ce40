"""Phase times of kpe_bs's workgroups (a library built with -DRMX_BS_STAMP, given as RMX_SO): where does a
workgroup's time go?  usage: RMX_SO=ab/bs_stamp.so RMX_KPE_BS=1 python tools/bs_stamps.py [n] [dens]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from remap_amd import lib as L
from remap_amd.synth import scroller_v1

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
dens = int(sys.argv[2]) if len(sys.argv) > 2 else 200
frames, _ = scroller_v1(n, dens=dens)
n, H, W = frames.shape
ctx = L.Context(W, H, kmax=2048)
d = torch.from_numpy(frames).cuda()
for _ in range(3):
    med, desc, kc = ctx.kpe(d)
ctx.sync()
ra = int(os.environ.get("RMX_BS_RA", "12")) // 6 * 6
gx = ((n + 2) // 3 + 3) // 4
gy = (H + 2 * ra - 1) // (2 * ra)
nb = gx * gy
buf = np.zeros((nb, 8), np.uint64)
rc = ctx.lib.rmx_debug_kpe_scratch(ctx.h, buf.ctypes.data, buf.nbytes)
assert rc == 0, rc
t = buf[:, :7].astype(np.int64)
ok = (t[:, 6] > 0)
t = t[ok]
names = ["stream", "count+scan", "list", "stage1 (test)", "reserve rows", "stage2 (emit)"]
dt = np.diff(t, axis=1)
tot = t[:, 6] - t[:, 0]
print(f"{ok.sum()} of {nb} workgroups with stamps; s_memtime ticks (100 MHz? see guide: tick = shader cycle)")
for i, nm in enumerate(names):
    print(f"  {nm:16s} mean {dt[:, i].mean():10.0f}  median {np.median(dt[:, i]):10.0f}  p90 {np.percentile(dt[:, i], 90):10.0f}  share {dt[:, i].sum() / tot.sum():.3f}")
print(f"  total            mean {tot.mean():10.0f}")
cand = (buf[ok][:, 7] & 0xffffffff).astype(np.int64)
print("candidates per workgroup: mean %.0f max %d" % (cand.mean(), cand.max()))
span = t[:, 6].max() - t[:, 0].min()
print("kernel span (ticks):", span, " sum of workgroup totals / span = average workgroups in flight:", tot.sum() / span)
