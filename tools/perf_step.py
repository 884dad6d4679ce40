"""Steps of the whole hot path (AlignerPipe, blocks in flight) on a scroller-v1 stream of a given density: ms per
step; under rocprofv3 --kernel-trace the timeline of one workload alone (tools/trace_timeline.py).
usage: python tools/perf_step.py [n] [dens] [lanes] [steps] [W H]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from remap_amd.pipeline import AlignerPipe
from remap_amd.synth import scroller_v1

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
dens = int(sys.argv[2]) if len(sys.argv) > 2 else 200
lanes = int(sys.argv[3]) if len(sys.argv) > 3 else 2
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
W, H = (int(sys.argv[5]), int(sys.argv[6])) if len(sys.argv) > 6 else (320, 240)
fr, _ = scroller_v1(n, W, H, dens=dens)
d = torch.from_numpy(fr).cuda()
pipe = AlignerPipe(W, H, kmax=4096 if W > 320 else 2048, lanes=lanes)
for _ in range(3 * lanes):
    pipe.align(d)
pipe.finish()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    out = pipe.align(d)
nf = pipe.finish()
torch.cuda.synchronize()
t1 = time.perf_counter()
print(f"n={n} dens={dens} lanes={lanes}: {1e3 * (t1 - t0) / steps:.3f} ms per step, {n * steps / (t1 - t0) / 1e6:.2f} M frames/s, "
      f"{float(out['kcount'].float().mean()):.1f} keypoints per frame, {nf} fragments, launch sets {[hex(a.ctx.kpm_launch_set()) for a in pipe.lanes]}")
pipe.close()
