"""BASELINE.json configs[4] (scaled): rooms-v1 top-down stream -> collect (kpe+kpm+frc+blit)
-> fgs splice (blend, kpe 1x1, all-pairs cellular match, greedy merges), all on one GPU.
Prints fragment statistics and timings; the 2 000-frame prefix is what the oracle can check."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from remap_amd.pipeline import Aligner
from remap_amd.splice import Fragment, splice
from remap_amd.synth import rooms_v1

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
rooms = int(sys.argv[2]) if len(sys.argv) > 2 else 8
W, H = 320, 240
frames, room, cam = rooms_v1(n, W, H, n_rooms=rooms, dens=200, seed=7)
al = Aligner(W, H, kmax=2048)
d = torch.from_numpy(frames).cuda()
torch.cuda.synchronize(); t0 = time.perf_counter()
out = al.align(d, blit=True)
torch.cuda.synchronize(); t_collect = time.perf_counter() - t0
pos = out["pos"].cpu().numpy(); frag = out["frag"].cpu().numpy()
frags = []
for fid in range(out["nfrag"]):
    canvas, zero, _ = out["canvases"][fid]
    idx = np.nonzero(frag == fid)[0]
    f = Fragment(canvas.clone(), zero, (W, H), [(int(i), pos[i]) for i in idx]); f.normalize(); frags.append(f)
dims = [f.dims for f in frags]
torch.cuda.synchronize(); t0 = time.perf_counter()
try:
    res = splice(frags)
    torch.cuda.synchronize(); t_splice = time.perf_counter() - t0
    print(json.dumps(dict(frames=n, rooms=rooms, collected=len(frags), spliced=len(res), collect_s=t_collect, splice_s=t_splice,
                          initial_pairs=len(frags) * (len(frags) - 1) // 2, max_dims=max(dims), out_dims=[f.dims for f in res][:8])))
except Exception as e:
    print(json.dumps(dict(frames=n, rooms=rooms, collected=len(frags), error=str(e)[:200], max_dims=max(dims))))
