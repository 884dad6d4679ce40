"""AlignerPipe.submit(blit=True) for a few blocks (for `rocprofv3 --kernel-trace`: which kernels
of the two lanes actually run beside each other)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from remap_amd.pipeline import AlignerPipe
from remap_amd.synth import scroller_v1

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
blit = (sys.argv[2] != "0") if len(sys.argv) > 2 else True
frames, _ = scroller_v1(n, dens=200)
d = torch.from_numpy(frames).cuda()
pipe = AlignerPipe(320, 240, kmax=2048, lanes=2)
for _ in range(12):
    pipe.submit(d, blit=blit)
pipe.drain()
torch.cuda.synchronize()
pipe.close()
