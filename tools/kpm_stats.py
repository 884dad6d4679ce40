"""Debug statistics of kpm_match (library built with RMX_CFLAGS=-DRMX_STATS): items reaching the
selection step per tier, how many take the exact tie path, mean distinct offsets / pairs."""
import sys, os, ctypes as C
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from remap_amd import lib as L
from remap_amd.synth import scroller_v1
dens = int(sys.argv[1]) if len(sys.argv) > 1 else 200
N = 10000
fr, _ = scroller_v1(N, 320, 240, dens=dens)
frames = torch.from_numpy(fr).cuda()
c = L.Context(320, 240, kmax=2048)
_, desc, kc = c.kpe(frames)
c.kpm_seq(desc, kc)
torch.cuda.synchronize()


print(c.kpm_status())
