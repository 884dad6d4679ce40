"""Per-launch duration of the extraction kernel with the lanes free-running (no turns) against taking turns.
usage: python tools/free_running_frac.py [lanes]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from remap_amd.pipeline import AlignerPipe
from remap_amd.synth import scroller_v1

lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 2
fr, _ = scroller_v1(10000, dens=200)
d = torch.from_numpy(fr).cuda()
alg = 2 * 320 * 240 * 10000 + 18 * 67.4554 * 10000
pipe = AlignerPipe(320, 240, kmax=2048, lanes=lanes)
if os.environ.get("PRELOAD"):  # unrelated device work first: is the slow first phase the device (clocks) or this pipeline's state?
    a = torch.randn(8192, 8192, device="cuda", dtype=torch.bfloat16)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < float(os.environ["PRELOAD"]):
        b = a @ a
        torch.cuda.synchronize()
for turns in [t == "1" for t in os.environ.get("SEQ", "1010")]:
    pipe.set_turns(turns)
    for _ in range(3 * lanes):
        pipe.align(d)
    pipe.finish()
    for lane in pipe.lanes:
        lane.ctx.set_timing(2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        pipe.align(d)
    pipe.finish()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    nl = [lane.ctx.kernel_launches("kpe_main") for lane in pipe.lanes]
    ms = sum(lane.ctx.kernel_ms("kpe_main") * k for lane, k in zip(pipe.lanes, nl)) / sum(nl)
    for lane in pipe.lanes:
        lane.ctx.set_timing(False)
    print(f"turns={turns}: {1e3 * dt:.3f} ms per step = {10000 / dt / 1e6:.2f} M frames/s; extraction kernel {ms:.3f} ms per launch = {alg / ms / 1e6 / 8000:.3f} of the peak by that duration; "
          f"{alg / (1e3 * dt) / 1e6 / 8000:.3f} by the step time")
pipe.close()
