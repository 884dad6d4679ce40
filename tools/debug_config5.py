"""All initial pair votes of the config-5 stream three ways: LDS instance, global-memory instance, oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import orclib as O
from remap_amd import lib as L
from remap_amd.pipeline import Aligner
from remap_amd.splice import Fragment, splice, _extract_single
from remap_amd.synth import rooms_v1

n, rooms = int(sys.argv[1]), int(sys.argv[2])
W, H = 320, 240
frames, room, cam = rooms_v1(n, W, H, n_rooms=rooms, dens=200, seed=7)
al = Aligner(W, H, kmax=2048)
out = al.align(torch.from_numpy(frames).cuda(), blit=True)
pos = out["pos"].cpu().numpy(); frag = out["frag"].cpu().numpy()
frags = []
for fid in range(out["nfrag"]):
    canvas, zero, _ = out["canvases"][fid]
    idx = np.nonzero(frag == fid)[0]
    f = Fragment(canvas.clone(), zero, (W, H), [(int(i), pos[i]) for i in idx]); f.normalize(); frags.append(f)
ctxs = {}
def cc(key):
    if key not in ctxs:
        w, h, kmax = (16, 8, 4096) if key == "any" else key
        ctxs[key] = L.Context(w, h, grid=(1, 1), overlap=0, kmax=kmax)
    return ctxs[key]
sn = [_extract_single(cc, f, 0) for f in frags]
pairs = [(i, j) for i in range(len(sn)) for j in range(i + 1, len(sn))]
m = cc("any")
args = ([s.table for s in sn], [s.k for s in sn], [s.mask for s in sn], [s.fragment.dims for s in sn])
t = time.time(); v_small = m.kpm_cell_pairs(*args, pairs); ts = time.time() - t
os.environ["RMX_FGS_BIG"] = "1"
t = time.time(); v_big = m.kpm_cell_pairs(*args, pairs); tb = time.time() - t
del os.environ["RMX_FGS_BIG"]
print("fragments", len(sn), "pairs", len(pairs), "kmax of snippets", max(s.k for s in sn), "small %.3fs big %.3fs" % (ts, tb))
diff = [q for q in range(len(pairs)) if tuple(v_small[q].tolist()) != tuple(v_big[q].tolist())]
print("small vs big mismatches:", len(diff), [(pairs[q], v_small[q], v_big[q]) for q in diff[:5]])
print("valid small", int((v_small["valid"] == 1).sum()), "valid big", int((v_big["valid"] == 1).sum()))
# oracle on the valid ones and a sample of the others
tabs = [np.ascontiguousarray(s.table[: s.k].cpu().numpy()).reshape(-1).view(O.DESC_DTYPE) for s in sn]
masks = [s.mask.cpu().numpy() for s in sn]
rng = np.random.default_rng(0)
check = set(np.nonzero(v_small["valid"] == 1)[0].tolist()) | set(np.nonzero(v_big["valid"] == 1)[0].tolist()) | set(rng.choice(len(pairs), 2000, replace=False).tolist())
bad = 0
for q in sorted(check):
    i, j = pairs[q]
    mm, ov = O.orc_kpm_cell(tabs[i], masks[i], tabs[j], masks[j].shape)
    for name, v in (("small", v_small[q]), ("big", v_big[q])):
        ok = int(v["valid"]) == mm and (not mm or (int(v["dx"]), int(v["dy"]), int(v["count"])) == (int(ov["dx"]), int(ov["dy"]), int(ov["count"])))
        if not ok:
            bad += 1
            if bad < 8: print("ORACLE MISMATCH", name, pairs[q], v, mm, ov)
print("checked", len(check), "pairs against the oracle:", bad, "mismatches")
t = time.time(); res = splice(frags); print("spliced", len(res), "in %.3fs" % (time.time() - t))
