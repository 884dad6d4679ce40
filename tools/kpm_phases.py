"""Phase times of kpm_build_medium (a library built with -DRMX_KPM_STAMP, RMX_SO): s_memtime ticks per phase
summed over the region waves, read from the context's status words 40..48.
usage: RMX_SO=ab/bs_kpmstamp.so python tools/kpm_phases.py [n] [dens]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from remap_amd import lib as L
from remap_amd.synth import scroller_v1

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
dens = int(sys.argv[2]) if len(sys.argv) > 2 else 40
fr, _ = scroller_v1(n, dens=dens)
ctx = L.Context(320, 240, kmax=2048)
med, desc, kc = ctx.kpe(torch.from_numpy(fr).cuda())
ctx.kpm_seq(desc, kc)
torch.cuda.synchronize()
buf = (C.c_uint32 * 64)()
rc = ctx.lib.rmx_debug_status(ctx.h, buf)
assert rc == 0
t = np.array(buf[40:49], dtype=np.float64)
names = ["entries", "grouped (hash)", "sorted", "groups numbered", "waited for other regions", "points and groups exported", "code index exported", "header", "map order"]
for nm, v in zip(names, t):
    print(f"{nm:28s} {v / t.sum() * 100:5.1f} %   {v / (8 * n) / 100:.2f} us per region wave")
