import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import orclib as O
from remap_amd import lib as L
from remap_amd.synth import scroller_v1
frames,_ = scroller_v1(3, dens=200)
n,H,W = frames.shape
ctx = L.Context(W,H,kmax=4096)
med,desc,kc = ctx.kpe(torch.from_numpy(frames).cuda())
print("status", ctx.kpe_status(), kc.cpu().numpy())
med = med.cpu().numpy()
for i in range(n):
    omed,odesc,k = O.orc_kpe(frames[i])
    bad = np.argwhere(med[i]!=omed)
    print("frame",i,"bad",len(bad), "k",k)
    if len(bad):
        print(bad[:20])
        ys,xs = bad[:,0],bad[:,1]
        print("x%4 hist", np.bincount(xs%4, minlength=4), "x range", xs.min(), xs.max(), "y range", ys.min(), ys.max())
        print("lane (x//4)%62 hist", np.bincount((xs//4)%62, minlength=62))
        y,x = bad[0]
        print("gpu", med[i][y-1:y+2, x-3:x+4]); print("orc", omed[y-1:y+2, x-3:x+4]); print("img", frames[i][y-2:y+3, x-3:x+4])
    tabs = L.desc_to_numpy(desc,kc)
    print("desc equal", tabs[i].tobytes()==odesc.tobytes(), len(tabs[i]), len(odesc))
