"""Timing of rmx_fdf_filter_batch on the bench stream with screen-space sprites."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from remap_amd.pipeline import Aligner
from remap_amd.fdf import filter_fragments
from remap_amd.synth import scroller_v1, with_sprites
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
dens = int(sys.argv[2]) if len(sys.argv) > 2 else 200
fr, _ = scroller_v1(N, 320, 240, dens=dens)
fr = with_sprites(fr, n_sprites=3, size=15, seed=1)
d = torch.from_numpy(fr).cuda()
al = Aligner(320, 240, kmax=2048)
out = al.align(d, blit=True)
frags = al.fragments(d, out)
print("fragments", len(frags), [f.dims for f in frags][:4])
res = filter_fragments(frags, d, out["median"], 320, 240, ctx=al.ctx)
torch.cuda.synchronize()
al.ctx.set_timing(True)
t0 = time.perf_counter()
for _ in range(3):
    res, info = filter_fragments(frags, d, out["median"], 320, 240, ctx=al.ctx, want_masks=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
ms = al.ctx.kernel_ms("fdf") * al.ctx.kernel_launches("fdf") / 3
mb = al.ctx.kernel_ms("fgm_blit") * al.ctx.kernel_launches("fgm_blit") / 3
nc = np.mean([v[1] for v in info.values()])
print(f"fdf labelling kernels {ms:.2f} ms, masked blit {mb:.2f} ms, wall {dt * 1e3:.1f} ms per {N} frames "
      f"({N / dt / 1e6:.2f} M frames/s), {nc:.1f} contours kept per frame")
