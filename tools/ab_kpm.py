"""The matcher chain alone on a stream (default: the dense one): ms per kpm_seq call and the context timers of its
kernels, under RMX_SO variants.  usage: ab_kpm.py [n] [dens] [W H kmax]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from remap_amd import lib as L
from remap_amd.synth import scroller_v1

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
dens = int(sys.argv[2]) if len(sys.argv) > 2 else 40
W, H, kmax = (int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])) if len(sys.argv) > 5 else (320, 240, 2048)
tag = os.path.basename(os.environ.get("RMX_SO", "default"))
fr, _ = scroller_v1(n, W, H, dens=dens)
ctx = L.Context(W, H, kmax=kmax)
d = torch.from_numpy(fr).cuda()
med, desc, kc = ctx.kpe(d)
for _ in range(3):
    ctx.kpm_seq(desc, kc)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(10):
    ctx.kpm_seq(desc, kc)
torch.cuda.synchronize()
ms = (time.perf_counter() - t) / 10 * 1e3
ctx.set_timing(True)
ctx.kpm_seq(desc, kc)
parts = {k: round(ctx.kernel_ms(k), 4) for k in ("kpm_build", "kpm_match")}
ctx.set_timing(False)
dxdy, matched = ctx.kpm_seq(desc, kc)
w = torch.arange(1, n + 1, device="cuda", dtype=torch.int64)
chk = int(((dxdy.to(torch.int64)[:, 0] * 7 + dxdy.to(torch.int64)[:, 1] * 13 + matched.to(torch.int64)) * w).sum().item())
print(f"{tag:16s} {n} frames {W}x{H} dens {dens}, {float(kc.float().mean()):.0f} keypoints per frame: kpm_seq {ms:.3f} ms per call, timers {parts}, offsets checksum {chk}", flush=True)
