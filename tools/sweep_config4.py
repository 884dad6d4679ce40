"""BASELINE.json configs[3]: 10 000 frames 640x480 (pixel-doubled scroller-v1), 1 GPU —
strip-length / occupancy sweep of kpe_main.  kpe_main stages nothing in LDS (reuse lives in
registers), so the knobs that exist are the strip length (rows per wave: halo re-reads vs
number of waves) and the batch size (waves in flight).  Prints one JSON line per point."""
import json, os, sys, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def run(rps, n):
    code = f"""
import sys, json, os
sys.path.insert(0, {os.path.dirname(os.path.dirname(os.path.abspath(__file__)))!r})
import numpy as np, torch
from remap_amd import lib as L
from remap_amd.synth import scroller_v1, upscale2
fr,_ = scroller_v1({n}//4+1, dens=100)
fr = upscale2(fr)
fr = np.ascontiguousarray(np.concatenate([fr]*4)[:{n}])
ctx = L.Context(640, 480, kmax=4096)
d = torch.from_numpy(fr).cuda(); med = torch.empty_like(d); desc,kc = ctx.alloc_tables({n})
ctx.kpe(d, med, desc, kc); ctx.sync(); ctx.set_timing(True)
for _ in range(5): ctx.kpe(d, med, desc, kc)
ms = ctx.kernel_ms('kpe_main'); mf = ctx.kernel_ms('kpe_finish'); K=int(kc.sum().item())
print(json.dumps(dict(rows_per_strip={rps}, frames={n}, kpe_main_ms=ms, kpe_finish_ms=mf, alg_GBs=(2*640*480*{n}+18*K)/ms/1e6, frac=(2*640*480*{n}+18*K)/ms/1e6/8000, kp_per_frame=K/{n})))
"""
    env = dict(os.environ, RMX_RPS=str(rps))
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
    print(out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-300:], flush=True)

for n in (2500,):
    for rps in (16, 32, 48, 80, 120, 240, 480):
        run(rps, n)
