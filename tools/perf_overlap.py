"""Experiment: overlap kpm (latency-bound) of sub-batch i with kpe (HBM-bound) of sub-batch i+1
on two HIP streams.  usage: perf_overlap.py [parts]"""
import sys, time, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from remap_amd import lib as L
from remap_amd.synth import scroller_v1

P = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dens = int(sys.argv[2]) if len(sys.argv) > 2 else 200
W, H, N = 320, 240, 10000
fr, truth = scroller_v1(N, W, H, dens=dens)
dev = torch.device("cuda", 0)
frames = torch.from_numpy(fr).to(dev)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
cA = L.Context(W, H, kmax=2048, stream=sA.cuda_stream)
cB = L.Context(W, H, kmax=2048, stream=sB.cuda_stream)
c0 = L.Context(W, H, kmax=2048)
median = torch.empty_like(frames)
desc, kcount = c0.alloc_tables(N)
torch.cuda.synchronize()

def serial():
    c0.kpe(frames, median, desc, kcount)
    dxdy, m = c0.kpm_seq(desc, kcount)
    return c0.frc_scan(dxdy, m)

def overlapped():
    cuts = [N * i // P for i in range(P + 1)]
    outs = []
    for i in range(P):
        a, b = cuts[i], cuts[i + 1]
        with torch.cuda.stream(sA):
            cA.kpe(frames[a:b], median[a:b], desc[a:b], kcount[a:b])
            ev = torch.cuda.Event(); ev.record(sA)
        with torch.cuda.stream(sB):
            sB.wait_event(ev)
            halo = (desc[a - 1], kcount[a - 1:a]) if a else None
            outs.append(cB.kpm_seq(desc[a:b], kcount[a:b], halo=halo))
    with torch.cuda.stream(sB):
        dxdy = torch.cat([o[0] for o in outs]); m = torch.cat([o[1] for o in outs])
        return cB.frc_scan(dxdy, m), dxdy, m

for f, name in ((serial, "serial"), (overlapped, f"overlap x{P}")):
    for _ in range(3): r = f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): r = f()
    torch.cuda.synchronize()
    print(name, (time.perf_counter() - t0) / 20 * 1e3, "ms/step")
(_, dxdy, m) = overlapped()
torch.cuda.synchronize()
d = dxdy.cpu().numpy(); mm = m.cpu().numpy()
print("exact", bool(mm[1:].all() and (d[1:] == truth[1:]).all()))
