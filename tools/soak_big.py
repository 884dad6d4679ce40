"""Randomised soak of the global-memory matchers against the oracle (run by hand on a GPU box):
cellular pairs with heavy ties, grid tickets of frames with huge regions."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import orclib as O
from remap_amd import lib as L

def table(rng, k, ncodes, w, h):
    pos = rng.choice(w * h, size=k, replace=False)
    x, y = pos // h, pos % h
    order = np.lexsort((y, x))
    codes = rng.integers(0, 256, size=(ncodes, 13), dtype=np.uint8)
    codes[:, 12] = (codes[:, 12] & 0xF0) | rng.integers(1, 3, size=ncodes, dtype=np.uint8)
    d = np.zeros(k, O.DESC_DTYPE)
    d["code"] = codes[rng.integers(0, ncodes, size=k)]
    d["region_mask"] = 1
    d["x"] = x[order]; d["y"] = y[order]
    return d

bad = 0
nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
os.environ["RMX_FGS_BIG"] = "1"
ctx = L.Context(16, 8, grid=(1, 1), overlap=0, kmax=4096)
for seed in range(nseeds):
    rng = np.random.default_rng(1000 + seed)
    w0, h0, w1, h1 = (int(v) for v in rng.integers(60, 400, size=4))
    kp, kc = int(rng.integers(1, min(3000, w0 * h0 // 3))), int(rng.integers(1, min(3000, w1 * h1 // 3)))
    ncodes = int(rng.choice([3, 20, 200, 5000]))
    tabs = [table(rng, kp, ncodes, w0, h0), table(rng, kc, ncodes, w1, h1)]
    masks = [(rng.random((h0, w0)) < 0.8).astype(np.uint8), (rng.random((h1, w1)) < 0.8).astype(np.uint8)]
    dt = [torch.from_numpy(t.view(np.uint8).reshape(-1, 18)).cuda() for t in tabs]
    dm = [torch.from_numpy(m).cuda() for m in masks]
    votes = ctx.kpm_cell_pairs(dt, [kp, kc], dm, [(w0, h0), (w1, h1)], [(0, 1), (1, 0)])
    for v, (i, j) in zip(votes, [(0, 1), (1, 0)]):
        m, ov = O.orc_kpm_cell(tabs[i], masks[i], tabs[j], masks[j].shape)
        ok = int(v["valid"]) == m and (not m or (int(v["dx"]), int(v["dy"]), int(v["count"])) == (int(ov["dx"]), int(ov["dy"]), int(ov["count"])))
        if not ok:
            bad += 1
            print("CELL MISMATCH seed", seed, (i, j), v, m, ov)
print("cell pairs:", 2 * nseeds, "cases,", bad, "mismatches")
ctx.close()
del os.environ["RMX_FGS_BIG"]

# grid tickets: frames whose regions hold thousands of entries with FEW distinct codes (many ties)
bad2 = 0
ctx = L.Context(320, 240, kmax=16384)
for seed in range(max(2, nseeds // 3)):
    rng = np.random.default_rng(2000 + seed)
    frames = np.full((4, 240, 320), 6, np.uint8)
    pal = rng.integers(0, 16, size=int(rng.choice([2, 3, 16])))
    for f in range(4):
        box = pal[rng.integers(0, len(pal), size=(120, 84))]
        frames[f, 6:126, 4:88] = np.roll(box, -f, axis=1) if seed % 2 == 0 else box
    d = torch.from_numpy(frames).cuda()
    med, desc, kcn = ctx.kpe(d)
    rc, mk = ctx.kpe_status()
    if rc:
        print("skip seed", seed, "k", mk); continue
    dxdy, matched = ctx.kpm_seq(desc, kcn)
    rc, fl = ctx.kpm_status()
    if rc and fl == 128:
        ctx.kpm_resolve(desc, kcn, dxdy, matched)
    elif rc:
        print("status", rc, fl); bad2 += 1; continue
    tickets = ctx.kpm_tickets(4)
    tabs = L.desc_to_numpy(desc, kcn)
    for q in range(1, 4):
        m, dd, t = O.orc_kpm_grid(tabs[q - 1], tabs[q])
        got = (int(matched[q].item()), tuple(int(v) for v in dxdy[q].cpu().numpy()))
        if got != (m, dd if m else (0, 0)):
            bad2 += 1; print("GRID MISMATCH seed", seed, q, got, (m, dd))
        for r in range(8):
            for j in range(3):
                a, b = tickets[q, r, j], t[r, j]
                if int(a["valid"]) != int(b["valid"]) or (int(b["valid"]) and (int(a["dx"]), int(a["dy"]), int(a["count"])) != (int(b["dx"]), int(b["dy"]), int(b["count"]))):
                    bad2 += 1; print("TICKET MISMATCH seed", seed, q, r, j, a, b)
    print("grid seed", seed, "k", kcn.cpu().numpy().tolist(), "flags", fl)
print("grid: mismatches", bad2)
ctx.close()
sys.exit(1 if bad or bad2 else 0)
