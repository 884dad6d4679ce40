"""Batch invariance of rmx_kpe_batch: a block of frames extracted alone must equal the same frames
extracted inside a longer batch (medians, counts, tables).  usage: cmp_kpe_batches.py [n] [dens]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from remap_amd import lib as L
from remap_amd.synth import scroller_v1

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
dens = int(sys.argv[2]) if len(sys.argv) > 2 else 200
frames, _ = scroller_v1(n, dens=dens)
n, H, W = frames.shape
d = torch.from_numpy(frames).cuda()
ctx = L.Context(W, H, kmax=2048)
med, desc, kc = ctx.kpe(d)
assert ctx.kpe_status()[0] == 0
M, D, K = med.cpu().numpy(), desc.cpu().numpy(), kc.cpu().numpy()
ok = True
for a, b in [(0, 2500), (2500, 5000), (5000, 7777), (7777, n), (1, 2), (3, 7), (9998, n)]:
    m2, d2, k2 = ctx.kpe(d[a:b])
    m2, d2, k2 = m2.cpu().numpy(), d2.cpu().numpy(), k2.cpu().numpy()
    bm = np.nonzero((m2 != M[a:b]).reshape(b - a, -1).any(1))[0]
    bk = np.nonzero(k2 != K[a:b])[0]
    bt = [i for i in range(b - a) if k2[i] == K[a + i] and d2[i, : k2[i]].tobytes() != D[a + i, : k2[i]].tobytes()]
    print(f"[{a},{b}) medians differ in {len(bm)} frames {bm[:6]}, counts in {len(bk)} {bk[:6]}, tables in {len(bt)} {bt[:6]}")
    for f in bm[:2]:
        ys, xs = np.nonzero(m2[f] != M[a + f])
        print("   frame", f, "rows", sorted(set(ys.tolist()))[:10], "cols", sorted(set(xs.tolist()))[:16], "n", len(ys), "sub", m2[f][ys[0], xs[0]], "whole", M[a + f][ys[0], xs[0]])
    for f in bk[:3]:
        print("   frame", f, "k sub", k2[f], "whole", K[a + f])
    ok = ok and not len(bm) and not len(bk) and not bt
print("OK" if ok else "MISMATCH")
