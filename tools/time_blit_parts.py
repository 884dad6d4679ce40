"""Where Aligner.blit spends its time on a stream with many fragments (rooms-v1): host preparation, the
zeroing of the stacked canvases, the launches.  usage: time_blit_parts.py [frames] [rooms]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from remap_amd.pipeline import Aligner
from remap_amd.synth import rooms_v1

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
rooms = int(sys.argv[2]) if len(sys.argv) > 2 else 16
frames, _, _ = rooms_v1(n, 320, 240, n_rooms=rooms, dens=200, seed=7)
al = Aligner(320, 240, kmax=2048)
d = torch.from_numpy(frames).cuda()
out = al.align(d)
sync = torch.cuda.synchronize
for rep in range(3):
    sync(); t0 = time.perf_counter()
    c = al.blit(d, out["pos"], out["frag"], 0, n)
    sync(); t1 = time.perf_counter()
    al.ctx.set_timing(True)
    al.blit(d, out["pos"], out["frag"], 0, n)
    ms, nl = al.ctx.kernel_ms("fgm_blit"), al.ctx.kernel_launches("fgm_blit")
    al.ctx.set_timing(False)
    print(f"rep {rep}: blit {1e3 * (t1 - t0):.2f} ms wall for {n} frames / {out['nfrag']} fragments; kernels {ms * nl:.2f} ms in {nl} launches; "
          f"flat buffer {al._layout[0].numel() * 2 / 1e6:.0f} MB")
sync(); t0 = time.perf_counter()
for _ in range(5):
    al.align(d)
sync(); ta = (time.perf_counter() - t0) / 5
sync(); t0 = time.perf_counter()
for _ in range(5):
    al.align(d, blit=True)
sync(); tb = (time.perf_counter() - t0) / 5
print(f"align {1e3 * ta:.2f} ms ({n / ta / 1e6:.2f} M frames/s), align + blit {1e3 * tb:.2f} ms ({n / tb / 1e6:.2f} M frames/s)")
