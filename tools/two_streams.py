"""Do the latency-bound kpm kernels of one batch hide under the VALU-bound kpe_mf of the next?
Steps enqueued back to back on ONE stream vs alternating over TWO contexts on two streams (two
independent streams of frames, each with its own workspace)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from remap_amd.pipeline import Aligner
from remap_amd.synth import scroller_v1

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
frames, truth = scroller_v1(n, dens=200)
d = torch.from_numpy(frames).cuda()
steps = 20

def run(nctx, timing=False, gate=False):
    streams = [torch.cuda.Stream() for _ in range(nctx)]
    als = []
    for s in streams:
        with torch.cuda.stream(s):
            als.append(Aligner(320, 240, kmax=2048))
    for i in range(2 * nctx):
        with torch.cuda.stream(streams[i % nctx]):
            als[i % nctx].align(d)
    torch.cuda.synchronize()
    for a in als:
        a.ctx.set_timing(2 if timing else 0)
    last = [None]
    def pre(s):
        if last[0] is not None:
            s.wait_event(last[0])
    def post(s):
        ev = torch.cuda.Event()
        ev.record(s)
        last[0] = ev
    t = time.perf_counter()
    for i in range(steps):
        s = streams[i % nctx]
        with torch.cuda.stream(s):
            out = als[i % nctx].align(d, defer=True, kpe_gate=(lambda s=s: pre(s), lambda s=s: post(s)) if gate else None)
    for a, s in zip(als, streams):
        with torch.cuda.stream(s):
            a.finish()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t) / steps * 1e3
    ok = bool(np.array_equal(out["dxdy"].cpu().numpy()[1:], truth[1:]))
    kpe = [a.ctx.kernel_ms("kpe_main") for a in als] if timing else []
    print(f"{nctx} context(s){' gated' if gate else ''}: {ms:.3f} ms per step, {n / ms / 1e3:.2f} M frames/s, offsets exact: {ok}" + (f", kpe_mf per launch {kpe}" if timing else ""), flush=True)
    for a in als:
        a.close()

for k in (1, 2, 3, 1, 2):
    run(k)
for k in (1, 2, 3):
    run(k, timing=True)
for k in (2, 3, 2, 3):
    run(k, gate=True)
for k in (2, 3):
    run(k, timing=True, gate=True)
