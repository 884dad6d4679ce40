import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from remap_amd.pipeline import Aligner
from remap_amd.synth import scroller_v1
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
frames, _ = scroller_v1(n, dens=200)
al = Aligner(320, 240)
d = torch.from_numpy(frames).cuda()
out = al.align(d, blit=True)
al.ctx.set_timing(True)
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    c = al.blit(d, out["pos"], out["frag"], 0, n)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    print("blit host+device ms", (t1 - t0) * 1e3, "kernel ms (4 kernels)", al.ctx.kernel_ms("fgm_blit"), "canvas", tuple(c[0][0].shape))
