"""fgm_blit under RMX_SO variants: (a) the bench stream's one canvas, adding to a canvas that holds counts and
writing a fresh one; (b) the pixel-doubled 640x480 stream (~1 000 fragments, ~10 GB of canvases per block)
through Aligner.blit.  Prints kernel time (context timers) and a checksum of the canvases (equal between
libraries).  usage: ab_blit.py [n] [what: ab]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from remap_amd import lib as L
from remap_amd.pipeline import Aligner
from remap_amd.synth import scroller_v1, upscale2

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
what = sys.argv[2] if len(sys.argv) > 2 else "ab"
sync = torch.cuda.synchronize
tag = os.path.basename(os.environ.get("RMX_SO", "default"))


def chk(t):
    v = t.reshape(-1).view(torch.int16).to(torch.int64)
    w = torch.arange(v.numel(), device=v.device, dtype=torch.int64) % 65521 + 1
    return int((v * w).sum().item() & 0xffffffffffff)


frames, truth = scroller_v1(n, dens=200)
if "a" in what:
    pos = np.cumsum(truth, axis=0).astype(np.int32)
    d = torch.from_numpy(frames).cuda()
    dp = torch.from_numpy(pos).cuda()
    ctx = L.Context(320, 240)
    cw, ch, zx, zy = ctx.fgm_extent(pos)
    canvas = torch.zeros((ch, cw, 16), dtype=torch.int16, device="cuda")
    for fresh in (False, True):
        for _ in range(2):
            ctx.fgm_blit(d, dp, canvas, (zx, zy), fresh=fresh)
        sync(); t = time.perf_counter()
        for _ in range(10):
            ctx.fgm_blit(d, dp, canvas, (zx, zy), fresh=fresh)
        sync(); ms = (time.perf_counter() - t) / 10 * 1e3
        ctx.set_timing(True)
        ctx.fgm_blit(d, dp, canvas, (zx, zy), fresh=fresh)
        kms = ctx.kernel_ms("fgm_blit")
        ctx.set_timing(False)
        canvas.zero_()
        ctx.fgm_blit(d, dp, canvas, (zx, zy), fresh=fresh)
        print(f"{tag:24s} scroller {'fresh' if fresh else 'add  '}: {ms:.3f} ms per call, timer {kms:.3f} ms, canvas {ch}x{cw}, checksum {chk(canvas):x}", flush=True)
    del d, canvas
if "b" in what:
    d = torch.empty((n, 480, 640), dtype=torch.uint8, device="cuda")
    for i in range(0, n, 1000):
        d[i:i + 1000] = torch.from_numpy(upscale2(frames[i:i + 1000])).cuda()
    al = Aligner(640, 480, kmax=4096)
    out = al.align(d)
    sync()
    c = al.blit(d, out["pos"], out["frag"], 0, n)
    sync(); t = time.perf_counter()
    for _ in range(3):
        c = al.blit(d, out["pos"], out["frag"], 0, n)
    sync(); ms = (time.perf_counter() - t) / 3 * 1e3
    al.ctx.set_timing(True)
    c = al.blit(d, out["pos"], out["frag"], 0, n)
    kms, nl = al.ctx.kernel_ms("fgm_blit"), al.ctx.kernel_launches("fgm_blit")
    al.ctx.set_timing(False)
    flat = al._layout[0]
    print(f"{tag:24s} doubled 640x480: blit {ms:.3f} ms per block of {n} ({out['nfrag']} fragments, {flat.numel() * 2 / 1e9:.2f} GB of canvases), "
          f"timer {kms * nl:.3f} ms in {nl} launches, checksum {chk(flat):x}", flush=True)
