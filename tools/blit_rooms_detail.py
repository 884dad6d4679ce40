"""Per-class detail of Aligner.blit on the rooms stream: stacked canvas sizes, frames and kernel time per launch.
usage: blit_rooms_detail.py [frames] [rooms]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from remap_amd.pipeline import Aligner, canvas_boxes
from remap_amd.synth import rooms_v1

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
rooms = int(sys.argv[2]) if len(sys.argv) > 2 else 64
frames, _, _ = rooms_v1(n, 320, 240, n_rooms=rooms, dens=200, seed=7)
al = Aligner(320, 240, kmax=2048)
d = torch.from_numpy(frames).cuda()
out = al.align(d)
pos = out["pos"].cpu().numpy().astype(np.int64); frag = out["frag"].cpu().numpy()
cuts = np.flatnonzero(frag[1:] != frag[:-1]) + 1
starts = np.concatenate([[0], cuts])
gw, gh, gzx, gzy = canvas_boxes(pos, starts, 320, 240)
lens = np.diff(np.concatenate([starts, [n]]))
heights, cls = np.unique(gh, return_inverse=True)
for c, h in enumerate(heights):
    m = cls == c
    wpad = (-(-gw[m] // 64) * 64)
    print(f"class {c}: height {h}, {m.sum()} fragments, stacked width {wpad.sum()}, dots {h * wpad.sum() / 1e6:.1f} M = {h * wpad.sum() * 32 / 1e9:.2f} GB, "
          f"{lens[m].sum()} frames, frame area / canvas area = {lens[m].sum() * 320 * 240 / (h * wpad.sum()):.1f}, widths {np.percentile(gw[m], [0, 50, 100]).astype(int).tolist()}")
c = al.blit(d, out["pos"], out["frag"], 0, n)
torch.cuda.synchronize()
al.ctx.set_timing(2)
al.blit(d, out["pos"], out["frag"], 0, n)
print("fgm_blit avg ms per launch", al.ctx.kernel_ms("fgm_blit"), "launches", al.ctx.kernel_launches("fgm_blit"))
