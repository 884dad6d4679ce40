"""BASELINE.json configs[3] end to end: 10 000 frames 640x480 (pixel-doubled scroller-v1), 1 GPU,
kpe + kpm + frc (+ blit, + codec).  Prints one JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from remap_amd.pipeline import Aligner
from remap_amd.synth import scroller_v1, upscale2

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
dens = int(sys.argv[2]) if len(sys.argv) > 2 else 200
fr, truth = scroller_v1(N, dens=dens)
d = torch.empty((N, 480, 640), dtype=torch.uint8, device="cuda")
for i in range(0, N, 1000):  # upscale in slices: the host copy of the doubled stream is 3 GB
    d[i:i + 1000] = torch.from_numpy(upscale2(fr[i:i + 1000])).cuda()
al = Aligner(640, 480, kmax=4096)
out = al.align(d)
dx = out["dxdy"].cpu().numpy(); mt = out["matched"].cpu().numpy()
exact = bool(mt[1:].all() and (dx[1:] == 2 * truth[1:]).all())
res = {}
for name, kw in (("align", {}), ("align+blit", dict(blit=True)), ("align+blit+codec", dict(blit=True, compress=True))):
    al.align(d, **kw); torch.cuda.synchronize()
    t0 = time.perf_counter(); steps = 5
    for _ in range(steps): al.align(d, **kw)
    torch.cuda.synchronize()
    res[name] = N * steps / (time.perf_counter() - t0)
al.ctx.set_timing(True)
for _ in range(3): al.align(d)
km = {k: al.ctx.kernel_ms(k) * al.ctx.kernel_launches(k) / 3 for k in ("kpe_main", "kpe_finish", "kpe_table", "kpm_build", "kpm_match")}
K = int(out["kcount"].sum().item())
alg = 2.0 * 640 * 480 * N + 18.0 * K
print(json.dumps(dict(workload=f"{N} frames 640x480 pixel-doubled scroller-v1 dens {dens}", offsets_exact=exact,
                      keypoints_per_frame=K / N, frames_per_s=res, kernel_ms_per_step=km,
                      kpe_main_alg_GBs=alg / km["kpe_main"] / 1e6, kpe_main_frac_of_8TBs=alg / km["kpe_main"] / 1e6 / 8000)))
