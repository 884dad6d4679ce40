"""Is kpe_main sensitive to where its buffers lie?  One big allocation; frames at offset 0, the
median plane at (size of frames + pad): time per launch for several pads, then the same with
fresh torch allocations."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from remap_amd import lib as L
from remap_amd.synth import scroller_v1
n = 10000
frames, _ = scroller_v1(n, dens=200)
ctx = L.Context(320, 240, kmax=2048)
src = torch.from_numpy(frames).cuda()
nbytes = src.numel()
desc, kc = ctx.alloc_tables(n)
pool = torch.empty(3 * nbytes + (64 << 20), dtype=torch.uint8, device="cuda")
def run(d, med, reps=20):
    ctx.kpe(d, med, desc, kc); ctx.sync()
    ctx.set_timing(True)
    for _ in range(reps):
        ctx.kpe(d, med, desc, kc)
    return ctx.kernel_ms("kpe_main"), ctx.kernel_ms("kpe_finish")
print("pool base %x" % pool.data_ptr())
for pad in (0, 256, 1024, 4096, 16384, 65536, 1 << 18, 1 << 20, (1 << 20) + 4096, 1 << 22, 3 << 20, 1 << 24, (1 << 24) + (1 << 19)):
    d = pool[:nbytes].view(n, 240, 320); d.copy_(src)
    off = nbytes + pad
    med = pool[off:off + nbytes].view(n, 240, 320)
    a, b = run(d, med)
    print(f"median at frames + {nbytes} + {pad:>9d}: kpe_main {a:.3f} ms  finish {b:.3f}")
for k in range(6):
    junk = torch.empty((k * 7919 + 1) * 4096, dtype=torch.uint8, device="cuda")
    d = src.clone(); med = torch.empty_like(d)
    a, b = run(d, med)
    print(f"fresh allocations {k}: frames %x median %x: kpe_main {a:.3f} ms  finish {b:.3f}" % (d.data_ptr(), med.data_ptr()))
    del junk
