"""How long does the HOST take to enqueue one step (all launches of AlignerPipe.align) against how long
the GPU takes to run it?  If the two are close, the step is launch-bound on the host.
usage: python tools/host_enqueue_time.py [n] [dens] [lanes]"""
import sys
import time

import os

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from remap_amd.pipeline import AlignerPipe
from remap_amd.synth import scroller_v1

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
dens = int(sys.argv[2]) if len(sys.argv) > 2 else 200
lanes = int(sys.argv[3]) if len(sys.argv) > 3 else 2
fr, _ = scroller_v1(n, dens=dens)
d = torch.from_numpy(fr).cuda()
pipe = AlignerPipe(320, 240, kmax=2048, lanes=lanes)
for _ in range(2 * lanes):
    pipe.align(d)
pipe.finish()
torch.cuda.synchronize()
K = 20
t0 = time.perf_counter()
for _ in range(K):
    pipe.align(d)
t1 = time.perf_counter()
pipe.finish()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"lanes={lanes}: host enqueue {1e3 * (t1 - t0) / K:.3f} ms per step; until the GPU is done {1e3 * (t2 - t0) / K:.3f} ms per step")
pipe.close()
