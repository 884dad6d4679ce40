"""Timing of rmx_nic_compress_batch on frames and medians of the bench stream."""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from remap_amd import lib as L
from remap_amd.synth import scroller_v1
N = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
dens = int(sys.argv[2]) if len(sys.argv) > 2 else 200
fr, _ = scroller_v1(N, 320, 240, dens=dens)
d = torch.from_numpy(fr).cuda()
c = L.Context(320, 240, kmax=2048)
med, _, _ = c.kpe(d)
for name, plane in (("frames", d), ("medians", med)):
    blobs, sizes = c.nic_compress(plane, pack=False)
    c.set_timing(True)
    for _ in range(5):
        c.nic_compress(plane, pack=False)
    c.sync()
    ms = c.kernel_ms("nic_compress")
    c.set_timing(False)
    tot = int(sizes.sum().item())
    print(f"{name}: {ms:.3f} ms per {N} images, {N / ms / 1e3:.2f} M images/s, in {plane.numel() / ms / 1e6:.0f} GB/s, "
          f"out {tot / N:.0f} B/image (ratio {plane.numel() / tot:.1f})")
# decode: rmx_nic_decompress_batch of the packed frame blobs
packed, sizes = c.nic_compress(d.reshape(N, -1))
back = c.nic_decompress(packed, sizes, 320 * 240)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    back = c.nic_decompress(packed, sizes, 320 * 240)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print(f"decompress: {dt * 1e3:.2f} ms per {N} images (wall, incl. offset upload), round trip exact: {bool(torch.equal(back, d.reshape(N, -1)))}")
