"""Where a step of tier B (align + canvas blit) spends its time: wall time per phase with a
device sync after each (so the phases do not overlap: an upper bound per phase)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from remap_amd.pipeline import Aligner
from remap_amd.synth import scroller_v1

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
frames, _ = scroller_v1(n, dens=200)
d = torch.from_numpy(frames).cuda()
al = Aligner(320, 240, kmax=2048)
for _ in range(3):
    out = al.align(d, blit=True)
torch.cuda.synchronize()

def T(fn, reps=10):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps):
        r = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps * 1e3, r

ms_align, out = T(lambda: al.align(d))
ms_full, _ = T(lambda: al.align(d, blit=True))
pos, frag = out["pos"], out["frag"]
ms_cpu, (ph, fh) = T(lambda: (pos.cpu().numpy(), frag.cpu().numpy()))
ms_uniq, _ = T(lambda: np.unique(fh))
ms_ext, ext = T(lambda: al.ctx.fgm_extent(ph))
ms_blit, _ = T(lambda: al.blit(d, pos, frag, 0, n))
c = next(iter(al.blit(d, pos, frag, 0, n).values()))[0]
ms_zero, _ = T(lambda: c.zero_())
ms_kernel, _ = T(lambda: al.ctx.fgm_blit(d, pos, c, (0, 0), frag_id=frag, fragment=0))
print(f"align {ms_align:.3f}  align+blit {ms_full:.3f}  | pos/frag to host {ms_cpu:.3f}  unique {ms_uniq:.3f}  extent {ms_ext:.3f}  "
      f"blit() whole {ms_blit:.3f}  canvas zero {ms_zero:.3f} ({c.numel()*2/1e6:.0f} MB)  fgm_blit call {ms_kernel:.3f}")
