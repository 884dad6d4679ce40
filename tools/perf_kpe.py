import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from remap_amd import lib as L
from remap_amd.synth import scroller_v1
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
dens = int(sys.argv[2]) if len(sys.argv) > 2 else 200
Wf = int(sys.argv[3]) if len(sys.argv) > 3 else 320  # (640 480: BASELINE configs[3], generated at that size)
Hf = int(sys.argv[4]) if len(sys.argv) > 4 else 240
frames,_ = scroller_v1(n, Wf, Hf, dens=dens)
n,H,W = frames.shape
ctx = L.Context(W,H,kmax=4096 if W > 320 else 2048)
d = torch.from_numpy(frames).cuda()
med = torch.empty_like(d); desc,kc = ctx.alloc_tables(n)
ctx.kpe(d, med, desc, kc); ctx.sync()
ctx.set_timing(True)
for it in range(3):
    ctx.set_timing(True); ctx.kpe(d, med, desc, kc)
    K = int(kc.sum().item())
    nl = ctx.kernel_launches("kpe_main")
    ms = ctx.kernel_ms("kpe_main") * nl; mt = ctx.kernel_ms("kpe_table"); mf = ctx.kernel_ms("kpe_finish") * nl
    byts = 2*W*H*n + 18*K
    print(f"launches={nl} n={n} dens={dens} K/frame={K/n:.1f} kpe_main {ms:.3f} ms -> {byts/ms/1e9*1e3/1e3:.3f} TB/s alg ({n/ms*1e3:.0f} fps) ; finish {mf:.3f} table {mt:.3f} ms")
ctx.set_timing(False)
torch.cuda.synchronize(); t=time.time()
for it in range(5): ctx.kpe(d, med, desc, kc)
torch.cuda.synchronize(); print("wall per call ms", (time.time()-t)/5*1e3)
