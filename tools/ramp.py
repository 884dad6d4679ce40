"""kpe_main's launch time over the first seconds of a process: does the device speed up under load?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from remap_amd import lib as L
from remap_amd.synth import scroller_v1
frames, _ = scroller_v1(10000, dens=200)
time.sleep(float(sys.argv[1]) if len(sys.argv) > 1 else 0.0)  # idle gap, as between generation and the first launch
ctx = L.Context(320, 240, kmax=2048)
d = torch.from_numpy(frames).cuda()
med = torch.empty_like(d); desc, kc = ctx.alloc_tables(10000)
t0 = time.perf_counter()
for it in range(40):
    ctx.set_timing(True)
    for _ in range(10 if it else 1):
        ctx.kpe(d, med, desc, kc)
    ms = ctx.kernel_ms("kpe_main"); mf = ctx.kernel_ms("kpe_finish")
    print(f"t={time.perf_counter()-t0:6.3f}s  kpe_main {ms:.3f} ms  finish {mf:.3f}")
