"""Same-box comparison of the extraction kernels on one stream: time per launch (HIP events), kernel names, and a
byte-for-byte comparison of medians and tables against the first variant.
usage: python tools/ab_pc.py [n] [dens] [variants...]   variant = name:ENV=V,ENV=V"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from remap_amd import lib as L
from remap_amd.synth import scroller_v1

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
dens = int(sys.argv[2]) if len(sys.argv) > 2 else 200
variants = sys.argv[3:] or ["ring:RMX_BS_RING=1,RMX_KPE_PC=0", "pc48:RMX_BS_RING=1", "pc24:RMX_BS_RING=1,RMX_PC_RA=24",
                            "pc96:RMX_BS_RING=1,RMX_PC_RA=96", "bs:RMX_BS_RING=0"]
frames, _ = scroller_v1(n, int(os.environ.get("ABW", "320")), int(os.environ.get("ABH", "240")), dens=dens)  # ABW / ABH: frame size
n, H, W = frames.shape
d = torch.from_numpy(frames).cuda()
ref = None
for rnd in range(2):
    for v in variants:
        name, _, envs = v.partition(":")
        keys = []
        for kv in filter(None, envs.split(",")):
            k, _, val = kv.partition("=")
            os.environ[k] = val
            keys.append(k)
        ctx = L.Context(W, H, kmax=2048 if W <= 320 else 4096)
        med = torch.empty_like(d)
        desc, kc = ctx.alloc_tables(n)
        ctx.kpe(d, med, desc, kc)
        ctx.sync()
        ts = []
        for it in range(4):
            ctx.set_timing(True)
            ctx.kpe(d, med, desc, kc)
            ctx.sync()
            ts.append(ctx.kernel_ms("kpe_main"))
        ctx.set_timing(False)
        K = int(kc.sum().item())
        byts = 2 * W * H * n + 18 * K
        best = min(ts)
        out = (med.cpu().numpy(), desc.cpu().numpy().copy(), kc.cpu().numpy().copy())
        same = ""
        if ref is None:
            ref = out
        elif rnd == 0:
            tabs_eq = all(np.array_equal(a, b) for a, b in zip(L.desc_to_numpy(desc, kc)[:200], L.desc_to_numpy(torch.from_numpy(ref[1]).cuda(), torch.from_numpy(ref[2]).cuda())[:200]))
            same = " medians %s counts %s tables[:200] %s" % (np.array_equal(out[0], ref[0]), np.array_equal(out[2], ref[2]), tabs_eq)
        print(f"{name:8s} {ctx.kpe_kernel():12s} K/frame {K / n:6.1f}  ms {best:.4f} (runs {' '.join('%.4f' % t for t in ts)})  {byts / best / 1e9:.3f} TB/s = {byts / best / 1e9 / 8:.3f} of 8{same}", flush=True)
        st = (C.c_uint32 * 64)()
        ctx.lib.rmx_debug_status(ctx.h, st)
        if st[40]:
            print(f"         stats: rolls {st[40]} candidates/roll {st[41] / st[40]:.1f} rounds/roll {st[42] / st[40]:.2f} rolls past the registers {st[43]}", flush=True)
        ctx.close()
        for k in keys:
            os.environ.pop(k, None)
