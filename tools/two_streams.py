"""Do the latency-bound kpm kernels of one block hide under the VALU-bound kpe_mf of the next?
One Aligner (steps enqueued back to back on one stream) against AlignerPipe with 2 / 3 lanes.
(Tried and dropped: the small kernels of a lane on a high-priority stream of their own — with
five streams on four hardware queues the steps got slower and erratic, 0.65-0.85 ms.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from remap_amd.pipeline import Aligner, AlignerPipe
from remap_amd.synth import scroller_v1

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
frames, truth = scroller_v1(n, dens=200)
d = torch.from_numpy(frames).cuda()
steps = 20

def run(lanes, timing=False):
    pipe = AlignerPipe(320, 240, kmax=2048, lanes=lanes)
    for _ in range(2 * lanes):
        pipe.align(d)
    pipe.finish()
    torch.cuda.synchronize()
    for a in pipe.lanes:
        a.ctx.set_timing(2 if timing else 0)
    t = time.perf_counter()
    for _ in range(steps):
        out = pipe.align(d)
    pipe.finish()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / steps * 1e3
    ok = bool(np.array_equal(out["dxdy"].cpu().numpy()[1:], truth[1:]))
    kpe = [round(a.ctx.kernel_ms("kpe_main"), 3) for a in pipe.lanes] if timing else ""
    print(f"{lanes} lane(s): {ms:.3f} ms per step, {n / ms / 1e3:.2f} M frames/s, offsets exact: {ok} {kpe}", flush=True)
    pipe.close()

for lanes in (1, 2, 3, 1, 2, 3):
    run(lanes)
for lanes in (1, 2, 3):
    run(lanes, timing=True)
