"""BASELINE.json configs[3]: 10 000 frames 640x480, 1 GPU — the LDS tile-size / occupancy sweep for the shipped
extraction kernels (kpe_pc, kpe_bs_ring).  Their "tile" is the strip a wave walks (rows per strip: halo rows and ring
restarts against the number of waves) and the workgroups a CU holds (LDS-limited: 38-40 KB per workgroup = 4; padded
to 3 and 2).  One JSON line per point: stream, kernel, rows per strip, workgroups per CU (= waves per SIMD), ms per
launch, algorithmic GB/s, fraction of the 8 TB/s peak, keypoints per frame.
usage: python tools/sweep_config4_r04.py [n] > profiles/r04_config4_strip_sweep_640x480.jsonl"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

from remap_amd import lib as L
from remap_amd.synth import scroller_v1, upscale2

N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
KNOBS = ("RMX_BS_RING", "RMX_KPE_PC", "RMX_PC_RA", "RMX_BS_RA", "RMX_PC_PADLDS", "RMX_BS_PADLDS")


def point(d, med, stream, kernel, ra, wgs):
    pad = {4: 0, 3: 13, 2: 40}[wgs]
    env = dict(RMX_BS_RING="1", RMX_KPE_PC="1" if kernel == "pc" else "0")
    env.update({"RMX_PC_RA": str(ra), "RMX_PC_PADLDS": str(pad)} if kernel == "pc" else {"RMX_BS_RA": str(ra), "RMX_BS_PADLDS": str(pad)})
    for k in KNOBS:
        os.environ.pop(k, None)
    os.environ.update(env)
    n = d.shape[0]
    ctx = L.Context(640, 480, kmax=4096)
    desc, kc = ctx.alloc_tables(n)
    ctx.kpe(d, med, desc, kc)
    ctx.sync()
    ts = []
    for _ in range(4):
        ctx.set_timing(True)
        ctx.kpe(d, med, desc, kc)
        ctx.sync()
        ts.append(ctx.kernel_ms("kpe_main"))
    K = int(kc.sum().item())
    ms = min(ts)
    alg = 2.0 * 640 * 480 * n + 18.0 * K
    print(json.dumps(dict(stream=stream, kernel=ctx.kpe_kernel(), rows_per_strip=ra, workgroups_per_cu=wgs, waves_per_simd=wgs, frames=n,
                          ms=ms, ms_runs=ts, alg_GBs=alg / ms / 1e6, frac=alg / ms / 1e6 / 8000, keypoints_per_frame=K / n)), flush=True)
    ctx.close()
    del desc, kc


for stream in ("native", "doubled"):
    if stream == "doubled":
        fr, _ = scroller_v1(N, dens=200)
        fr = upscale2(fr)
    else:
        fr, _ = scroller_v1(N, 640, 480, dens=200)
    d = torch.from_numpy(np.ascontiguousarray(fr)).cuda()
    del fr
    med = torch.empty_like(d)
    for kernel, ras in (("pc", (6, 12, 18, 24, 30, 36, 48, 60, 96)), ("ring", (6, 12, 18, 24, 30))):
        for ra in ras:
            for wgs in ((4, 3, 2) if ra in (12, 30) or (kernel == "ring" and ra == 24) else (4,)):
                point(d, med, stream, kernel, ra, wgs)
    del d, med
    torch.cuda.empty_cache()
