"""BASELINE.json's full size (10 000 frames 320x240) through size-independent properties:
ground-truth offsets of the generator, batch invariance (one batch == four batches chained
by halo tables and the scan carry), and oracle spot checks on sampled frames."""
import numpy as np
import pytest

import orclib as O
from remap_amd import lib as L
from remap_amd.synth import scroller_v1

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dens", [200, 40])
def test_ten_thousand_frames(gpu, dens):
    import torch

    n = 10000
    frames, truth = scroller_v1(n, dens=dens)
    ctx = L.Context(320, 240, kmax=2048)
    d = torch.from_numpy(frames).cuda()
    med, desc, kc = ctx.kpe(d)
    assert ctx.kpe_status()[0] == 0
    dxdy, matched = ctx.kpm_seq(desc, kc)
    assert ctx.kpm_status()[0] == 0
    pos, frag, carry = ctx.frc_scan(dxdy, matched)
    dx = dxdy.cpu().numpy()
    assert matched.cpu().numpy()[1:].all() and np.array_equal(dx[1:], truth[1:])
    assert carry[1] == 0 and np.array_equal(pos.cpu().numpy()[:, 0], np.cumsum(truth[:, 0]))
    # batch invariance: four chunks, each matched against the previous chunk's last table
    cuts = [0, 2500, 5000, 7777, n]
    parts_d, parts_m, parts_p = [], [], []
    cr = None
    for a, b in zip(cuts[:-1], cuts[1:]):
        m2, d2, k2 = ctx.kpe(d[a:b])
        halo = None if a == 0 else (desc[a - 1], kc[a - 1:a])
        dd, mm = ctx.kpm_seq(d2, k2, halo=halo)
        pp, ff, cr = ctx.frc_scan(dd, mm, first_is_start=(a == 0), carry=cr)
        assert torch.equal(m2, med[a:b]) and torch.equal(k2, kc[a:b])
        parts_d.append(dd)
        parts_m.append(mm)
        parts_p.append(pp)
    assert torch.equal(torch.cat(parts_d), dxdy) and torch.equal(torch.cat(parts_m), matched)
    assert torch.equal(torch.cat(parts_p), pos) and cr == carry
    # oracle spot checks
    tabs_k = kc.cpu().numpy()
    for i in (0, 1, 4999, 9999):
        omed, odesc, k = O.orc_kpe(frames[i])
        assert k == tabs_k[i] and np.array_equal(med[i].cpu().numpy(), omed)
        assert desc[i, :k].cpu().numpy().tobytes() == odesc.tobytes()
    ctx.close()


def test_deferred_steps_equal_synchronous_ones(gpu):
    """Aligner.align(defer=True): steps enqueued back to back, settled one step late (or in finish()) —
    same offsets, positions, fragment ids and fragment count as the synchronous step; an overflow in a
    deferred step surfaces when that step is settled: before the next step reuses its buffers."""
    import torch

    from remap_amd.pipeline import Aligner

    frames, truth = scroller_v1(600, dens=100)
    frames = frames.copy()
    frames[200] = 6  # fragment breaks
    frames[433] = 6
    d = torch.from_numpy(frames).cuda()
    al = Aligner(320, 240, kmax=2048)
    ref = al.align(d)
    want = {k: ref[k].clone() for k in ("dxdy", "matched", "pos", "frag", "kcount")}
    for _ in range(3):
        out = al.align(d, defer=True)
        assert out["nfrag"] is None
    assert al.finish() == ref["nfrag"]
    for k, v in want.items():
        assert torch.equal(out[k], v), k
    al.close()
    # a table overflow in a deferred step is reported for THAT step: when the next one starts, or in finish()
    rng = np.random.default_rng(1)
    noise = torch.from_numpy(rng.integers(0, 16, size=(4, 240, 320), dtype=np.uint8)).cuda()
    al = Aligner(320, 240, kmax=512)  # (the scroller's frames have ~270 keypoints, the noise 46 000)
    al.align(noise, defer=True)
    with pytest.raises(L.RemapError):
        al.align(d[:4].contiguous(), defer=True)
    assert al.align(d[:4].contiguous(), defer=True)["nfrag"] is None and al.finish() == 1  # (the Aligner goes on)
    al.align(noise, defer=True)
    with pytest.raises(L.RemapError):
        al.finish()
    al.close()


def test_blocks_in_flight_equal_blocks_one_by_one(gpu):
    """AlignerPipe: different blocks taken in turn by two (three) lanes on their own streams, the
    extraction kernels gated by events — every block's outputs equal the synchronous Aligner's;
    finish() reports the last block's fragment count and an overflow of ANY block."""
    import torch

    from remap_amd.pipeline import Aligner, AlignerPipe

    blocks = []
    for k, (n, dens) in enumerate(((300, 100), (500, 200), (200, 60), (400, 150), (350, 100))):
        fr, _ = scroller_v1(n, dens=dens, cam_seed=100 + k)
        fr = fr.copy()
        if k % 2:
            fr[n // 3] = 6  # a fragment break
        blocks.append(torch.from_numpy(fr).cuda())
    al = Aligner(320, 240, kmax=2048)
    want = []
    for b in blocks:
        o = al.align(b)
        want.append(({k: o[k].clone() for k in ("dxdy", "matched", "pos", "frag", "kcount", "median")}, o["nfrag"]))
    al.close()
    for lanes, depth in ((2, 1), (3, 1), (1, 1), (2, 2)):  # (depth: buffer sets per stream, taken in turn)
        pipe = AlignerPipe(320, 240, kmax=2048, lanes=lanes, depth=depth)
        slots = lanes * depth
        for rep in range(2):
            outs = [pipe.align(b) for b in blocks[:slots]]  # (a slot's buffers are reused by its next block)
            assert pipe.finish() == want[slots - 1][1]
            for o, (w, _) in zip(outs, want):
                for k, v in w.items():
                    assert torch.equal(o[k], v), (lanes, depth, k)
        # more blocks than lanes: each block is settled and read before its lane is used again
        for i, b in enumerate(blocks):
            o = pipe.align(b)
            assert pipe.settle(o) == want[i][1]
            for k, v in want[i][0].items():
                assert torch.equal(o[k], v), (lanes, i, k)
        assert pipe.finish() == want[-1][1]
        pipe.close()
    rng = np.random.default_rng(1)
    noise = torch.from_numpy(rng.integers(0, 16, size=(4, 240, 320), dtype=np.uint8)).cuda()
    pipe = AlignerPipe(320, 240, kmax=512, lanes=2)
    pipe.align(noise)
    pipe.align(blocks[0][:4].contiguous())
    with pytest.raises(L.RemapError):  # lane 0 is used again: its block is settled first
        pipe.align(blocks[0][:4].contiguous())
    pipe.align(noise)
    with pytest.raises(L.RemapError):  # ... or in finish(), after every lane has been joined
        pipe.finish()
    pipe.close()


def test_submit_drain_blits_equal_synchronous_steps(gpu):
    """AlignerPipe.submit / drain: blocks with a canvas blit (and the codec) in flight on two
    lanes — every block's canvases, blobs and fragment count equal the synchronous Aligner's."""
    import torch

    from remap_amd.pipeline import Aligner, AlignerPipe

    blocks = []
    for k, (n, dens) in enumerate(((300, 100), (260, 200), (200, 60), (320, 150))):
        fr, _ = scroller_v1(n, dens=dens, cam_seed=200 + k)
        fr = fr.copy()
        if k % 2 == 0:
            fr[n // 2] = 6  # a fragment break: two canvases
        blocks.append(torch.from_numpy(fr).cuda())
    al = Aligner(320, 240, kmax=2048)
    want = []
    for b in blocks:
        o = al.align(b, blit=True, compress=True)
        want.append(dict(nfrag=o["nfrag"], pos=o["pos"].clone(),
                         canv={f: (c.clone(), z, g) for f, (c, z, g) in o["canvases"].items()},
                         sizes=o["nic_image"][1].clone(), msizes=o["nic_median"][1].clone(), blobs=o["nic_median"][0].clone()))
    al.close()
    pipe = AlignerPipe(320, 240, kmax=2048, lanes=2)
    got = []

    def keep(r):  # (a lane's buffers are reused by its next block: compare right away)
        if r is None:
            return
        w = want[len(got)]
        assert r["nfrag"] == w["nfrag"] and torch.equal(r["pos"], w["pos"])
        assert set(r["canvases"]) == set(w["canv"])
        for f, (c, z, g) in r["canvases"].items():
            assert z == w["canv"][f][1] and g == w["canv"][f][2] and torch.equal(c, w["canv"][f][0]), f
        assert torch.equal(r["nic_image"][1], w["sizes"]) and torch.equal(r["nic_median"][1], w["msizes"])
        valid = torch.arange(w["blobs"].shape[1], device="cuda")[None, :] < w["msizes"][:, None]  # (bytes past a blob's size are padding)
        assert torch.equal(r["nic_median"][0] * valid, w["blobs"] * valid)
        got.append(True)

    for b in blocks:
        keep(pipe.submit(b, blit=True, compress=True))
        torch.cuda.synchronize()
    for r in pipe.drain():
        keep(r)
    assert len(got) == len(blocks)
    pipe.close()


def test_deferred_results_are_sealed_until_settled(gpu):
    """A deferred step's offsets, positions and fragment ids may still be rewritten when the step is settled
    (tickets left to rmx_kpm_resolve, a matcher instance outside the lane's launch set): reading them after a bare
    synchronize() raises; after settle() / finish() they are the synchronous step's.  The extraction's outputs are
    final at once."""
    import torch

    from remap_amd.pipeline import Aligner, AlignerPipe

    fr, _ = scroller_v1(300, dens=100)
    d = torch.from_numpy(fr).cuda()
    al = Aligner(320, 240, kmax=2048)
    ref = al.align(d)
    want = {k: ref[k].clone() for k in ("dxdy", "matched", "pos", "frag", "kcount", "median")}
    out = al.align(d, defer=True)
    torch.cuda.synchronize()
    assert out["nfrag"] is None and not out.settled
    for k in ("dxdy", "matched", "pos", "frag"):
        with pytest.raises(RuntimeError):
            out[k]
        with pytest.raises(RuntimeError):
            out.get(k)
    assert torch.equal(out["kcount"], want["kcount"]) and torch.equal(out["median"], want["median"])
    assert al.finish() == ref["nfrag"] and out.settled
    for k, v in want.items():
        assert torch.equal(out[k], v), k
    al.close()
    pipe = AlignerPipe(320, 240, kmax=2048, lanes=2)
    o1, o2 = pipe.align(d), pipe.align(d)
    torch.cuda.synchronize()
    for o in (o1, o2):
        with pytest.raises(RuntimeError):
            o["dxdy"]
    assert pipe.settle(o1) == ref["nfrag"] and o1.settled and not o2.settled
    assert torch.equal(o1["pos"], want["pos"])
    with pytest.raises(RuntimeError):
        o2["pos"]
    o3 = pipe.align(d)  # lane 0 again: nothing of o1 is touched before ... it was settled
    assert pipe.finish() == ref["nfrag"] and o2.settled and o3.settled
    assert torch.equal(o2["dxdy"], want["dxdy"]) and torch.equal(o3["frag"], want["frag"])
    pipe.close()


def test_bin_sum_widening_is_exact_on_device(gpu):
    """sum_bins_mod_2_16 (the canvases' cross-rank sum: int16 bins widened to int32 for the all-reduce, narrowed
    again) on DEVICE tensors with one rank: the identity for every uint16 value, wrapped ones included — the
    arithmetic of the nccl branch without a second GPU."""
    import torch

    from remap_amd.pipeline import sum_bins_mod_2_16

    v = torch.arange(0, 65536, dtype=torch.int32).to(torch.int16).cuda()
    flat = v.repeat(3).contiguous()
    before = flat.clone()
    assert sum_bins_mod_2_16(flat) is flat and torch.equal(flat, before)
    # two "ranks" by hand: the int32 sum of two uint16 values narrowed = their sum modulo 2^16
    a = torch.randint(-32768, 32768, (4096,), dtype=torch.int16, device="cuda")
    b = torch.randint(-32768, 32768, (4096,), dtype=torch.int16, device="cuda")
    wide = (a.to(torch.int32) & 0xffff) + (b.to(torch.int32) & 0xffff)
    got = (wide & 0xffff).to(torch.int16)
    want = ((a.cpu().numpy().view(np.uint16).astype(np.uint32) + b.cpu().numpy().view(np.uint16)) & 0xffff).astype(np.uint16)
    assert np.array_equal(got.cpu().numpy().view(np.uint16), want)
