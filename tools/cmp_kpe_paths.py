"""Same frames through kpe_mf and kpe_bs (RMX_KPE_BS is read when a context is created): medians, counts
and descriptor tables must be identical.  usage: cmp_kpe_paths.py [n] [dens] [W H]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from remap_amd import lib as L
from remap_amd.synth import scroller_v1, upscale2

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2500
dens = int(sys.argv[2]) if len(sys.argv) > 2 else 200
frames, _ = scroller_v1(n, dens=dens)
if len(sys.argv) > 3 and sys.argv[3] == "640":
    frames = upscale2(frames)
n, H, W = frames.shape
d = torch.from_numpy(frames).cuda()
res = {}
for mode in ("0", "1"):
    os.environ["RMX_KPE_BS"] = mode
    ctx = L.Context(W, H, kmax=4096)
    med, desc, kc = ctx.kpe(d)
    assert ctx.kpe_status()[0] == 0
    res[mode] = (med.cpu().numpy(), desc.cpu().numpy().copy(), kc.cpu().numpy())
    ctx.close()
m0, d0, k0 = res["0"]
m1, d1, k1 = res["1"]
bad = np.nonzero((m0 != m1).reshape(n, -1).any(1))[0]
print("frames with differing medians:", len(bad), bad[:10])
for f in bad[:3]:
    ys, xs = np.nonzero(m0[f] != m1[f])
    print(" frame", f, "rows", sorted(set(ys.tolist()))[:12], "cols", sorted(set(xs.tolist()))[:12], "count", len(ys))
badk = np.nonzero(k0 != k1)[0]
print("frames with differing counts:", len(badk), badk[:10], k0[badk[:10]], k1[badk[:10]])
same = all(d0[i, : k0[i]].tobytes() == d1[i, : k1[i]].tobytes() for i in range(n) if k0[i] == k1[i])
print("tables equal where counts agree:", same)
print("OK" if len(bad) == 0 and len(badk) == 0 and same else "MISMATCH")
