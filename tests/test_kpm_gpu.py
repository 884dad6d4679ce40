"""kpm / frc parity on the GPU: offsets, tickets (incl. libstdc++ tie order), positions."""
import numpy as np
import pytest

import orclib as O
from remap_amd import lib as L
from remap_amd.synth import scroller_v1

pytestmark = pytest.mark.gpu


def gpu_pipeline(frames, kmax=2048, split=None):
    import torch

    n, H, W = frames.shape
    ctx = L.Context(W, H, kmax=kmax)
    d = torch.from_numpy(frames).cuda()
    med, desc, kc = ctx.kpe(d)
    assert ctx.kpe_status()[0] == 0
    if split is None:
        dxdy, matched = ctx.kpm_seq(desc, kc)
        tickets = ctx.kpm_tickets(n)
    else:
        a = ctx.kpm_seq(desc[:split], kc[:split])
        ta = ctx.kpm_tickets(split)
        b = ctx.kpm_seq(desc[split:], kc[split:], halo=(desc[split - 1], kc[split - 1:split]))
        tb = ctx.kpm_tickets(n - split)
        dxdy = torch.cat([a[0], b[0]])
        matched = torch.cat([a[1], b[1]])
        tickets = np.concatenate([ta, tb])
    assert ctx.kpm_status()[0] == 0
    tabs = L.desc_to_numpy(desc, kc)
    out = dict(dxdy=dxdy.cpu().numpy(), matched=matched.cpu().numpy(), tickets=tickets, tabs=tabs, ctx=ctx,
               dev=(dxdy, matched))
    return out


def check_seq(frames, **kw):
    g = gpu_pipeline(frames, **kw)
    n = len(frames)
    ties = 0
    for i in range(1, n):
        m, d, t = O.orc_kpm_grid(g["tabs"][i - 1], g["tabs"][i])
        gt = g["tickets"][i]
        assert gt.tobytes() == t.tobytes(), f"tickets differ at frame {i}:\n{gt}\n{t}"
        assert g["matched"][i] == m, f"matched differs at frame {i}"
        assert tuple(g["dxdy"][i]) == d, f"offset differs at frame {i}: {g['dxdy'][i]} vs {d}"
        for r in range(8):
            c = [int(v["count"]) for v in t[r] if v["valid"]]
            ties += any(c[j] == c[j + 1] for j in range(len(c) - 1))
    assert g["matched"][0] == 0
    return g, ties


@pytest.mark.parametrize("dens", [200, 40])
def test_kpm_scroller(gpu, dens):
    frames, truth = scroller_v1(40, dens=dens)
    g, ties = check_seq(frames)
    assert (g["dxdy"][1:] == truth[1:]).all() and g["matched"][1:].all()
    if dens == 40:
        assert ties > 0  # the dense stream exercises the unordered_map tie order


def test_kpm_repetitive_tiles_many_ties(gpu):
    # a tiled world: many identical codes, large all-pairs groups, ambiguous offsets
    rng = np.random.default_rng(11)
    tile = rng.integers(0, 16, size=(16, 16), dtype=np.uint8)
    tile = np.kron(tile[:4, :4], np.ones((4, 4), np.uint8))
    world = np.tile(tile, (240 // 16 + 1, 40))[:240]
    world[100:140, 300:340] = 3  # one unique landmark
    frames = np.stack([world[:, x:x + 320] for x in (0, 3, 7, 8, 12, 16, 16, 21)])
    check_seq(np.ascontiguousarray(frames))


def test_kpm_sparse_frames_with_ambiguous_stamps(gpu):
    # few keypoints per frame (the pair-wave instance), plus one identical 4x4 stamp at random
    # places: code groups whose all-pairs offsets are mostly distinct -> equal counts inside
    # a region (tie order of the totalizator); every 6th frame carries many stamps, more
    # distinct offsets than the pair-wave tables hold (hand-over to the per-region tiers)
    frames, _ = scroller_v1(36, dens=300)
    frames = frames.copy()
    rng = np.random.default_rng(5)
    patch = rng.integers(0, 16, size=(4, 4), dtype=np.uint8)
    for i in range(len(frames)):
        for _ in range(int(rng.integers(2, 6)) if i % 6 else 14):
            y, x = int(rng.integers(8, 224)), int(rng.integers(8, 306))
            frames[i, y:y + 4, x:x + 4] = patch
    g, ties = check_seq(frames)
    assert ties > 10


def test_kpm_breaks_and_blank_frames(gpu):
    frames, _ = scroller_v1(12, dens=100)
    frames = frames.copy()
    frames[4] = 6            # blank frame: no keypoints -> active < 2 -> no match, and next too
    frames[8] = frames[8][::-1].copy()  # unrelated content
    g, _ = check_seq(frames)
    assert g["matched"][4] == 0 and g["matched"][5] == 0


def test_kpm_halo_equals_single_batch(gpu):
    frames, _ = scroller_v1(30, dens=40)
    a, _ = check_seq(frames)
    b = gpu_pipeline(frames, split=13)
    assert np.array_equal(a["dxdy"][1:], b["dxdy"][1:]) and np.array_equal(a["matched"][1:], b["matched"][1:])
    assert a["tickets"][1:].tobytes() == b["tickets"][1:].tobytes()


def test_frc_scan_matches_oracle(gpu):
    import torch

    frames, _ = scroller_v1(60, dens=100)
    frames = frames.copy()
    frames[20] = 6
    frames[41] = 6
    g = gpu_pipeline(frames)
    ctx = g["ctx"]
    dxdy, matched = g["dev"]
    pos, frag, carry = ctx.frc_scan(dxdy, matched)
    o = O.orc_frc(frames)
    assert np.array_equal(pos.cpu().numpy(), o["pos"])
    assert np.array_equal(frag.cpu().numpy().astype(np.uint32), o["frag"])
    assert np.array_equal(g["matched"], o["matched"]) and np.array_equal(g["dxdy"], o["dxdy"])
    # the same in two batches with the carry
    p1, f1, c1 = ctx.frc_scan(dxdy[:25], matched[:25])
    p2, f2, c2 = ctx.frc_scan(dxdy[25:], matched[25:], first_is_start=False, carry=c1)
    assert np.array_equal(torch.cat([p1, p2]).cpu().numpy(), o["pos"])
    assert np.array_equal(torch.cat([f1, f2]).cpu().numpy().astype(np.uint32), o["frag"])
    assert c2 == carry


def test_frc_scan_long_streams_random_breaks(gpu):
    """rmx_frc_scan on synthetic offsets: lengths around the 16 384-frame span of the kernel,
    break densities from none to every third frame, with and without a carry."""
    import torch

    from test_pipeline_gloo import _scan

    ctx = L.Context(320, 240, kmax=64)
    rng = np.random.default_rng(17)
    for n in (1, 2, 1023, 1025, 16383, 16384, 16385, 40000):
        for p_break in (0.0, 0.001, 0.3):
            matched = (rng.random(n) >= p_break).astype(np.uint8)
            dxdy = rng.integers(-5, 6, size=(n, 2)).astype(np.int32)
            first = bool(rng.integers(0, 2))
            carry = ((0, 0), 0) if first else ((int(rng.integers(-99, 99)), int(rng.integers(-99, 99))), int(rng.integers(0, 9)))
            pos, frag, cout = ctx.frc_scan(torch.from_numpy(dxdy).cuda(), torch.from_numpy(matched).cuda(),
                                           first_is_start=first, carry=None if first else carry)
            wp, wf, wc = _scan(dxdy, matched, first, carry)
            assert np.array_equal(pos.cpu().numpy(), wp) and np.array_equal(frag.cpu().numpy(), wf), (n, p_break)
            assert cout == wc, (n, p_break)
    ctx.close()


def test_kpm_large_tables_kmax_8192(gpu):
    # frames with thousands of keypoints (noise blocks) and a table stride above 2048
    rng = np.random.default_rng(21)
    base = np.kron(rng.integers(0, 16, size=(30, 50), dtype=np.uint8), np.ones((8, 8), np.uint8))
    for x0 in (40, 210):  # two noisy bands in different column regions (~2.8k keypoints per frame)
        base[:, x0:x0 + 10] = rng.integers(0, 16, size=(240, 10), dtype=np.uint8)
    frames = np.stack([np.ascontiguousarray(base[:, x:x + 320]) for x in (0, 2, 5, 5, 9)])
    g, _ = check_seq(frames, kmax=8192)
    assert max(len(t) for t in g["tabs"]) > 2048


@pytest.mark.parametrize("dens", [150, 120, 90, 70, 55])
def test_kpm_build_tiers_between_sparse_and_dense(gpu, dens):
    """Densities at which regions hold 10-30, 17-32, 33-64 and 65-256 entries: the half-wave
    first pass, its 32-key sorts, the wave-per-region instance (tier 4) and the first medium
    instance all build tables the matcher reads bit-for-bit like the oracle."""
    frames, _ = scroller_v1(14, dens=dens)
    g, _ = check_seq(frames)
    per_region = [np.bincount(np.concatenate([np.flatnonzero([(m >> r) & 1 for m in t["region_mask"]]) * 0 + r for r in range(8)]),
                              minlength=8).max() for t in g["tabs"]]
    assert max(per_region) > 16  # beyond the 16-key network


def test_frc_block_scan_equals_whole_stream_scan(gpu):
    """rmx_frc_scan_block + rmx_frc_apply_blocks (the frame-sharded chain without host round
    trips) over 1..5 blocks of a stream with breaks = rmx_frc_scan over the whole stream."""
    import torch

    rng = np.random.default_rng(3)
    ctx = L.Context(320, 240)
    for trial in range(6):
        n = int(rng.integers(40, 4000))
        dxdy = torch.from_numpy(rng.integers(-4, 5, size=(n, 2)).astype(np.int32)).cuda()
        m = (rng.random(n) > (0.0, 0.02, 0.3, 0.9, 1.0, 0.5)[trial]).astype(np.uint8)
        if trial == 5:
            m[: n // 2] = 1  # no break in the first blocks: carries run across several of them
        matched = torch.from_numpy(m).cuda()
        pos, frag, carry = ctx.frc_scan(dxdy, matched, first_is_start=True)
        for world in (1, 2, 3, 5):
            cuts = [0] + sorted(rng.choice(np.arange(1, n), size=world - 1, replace=False).tolist()) + [n]
            parts = [ctx.frc_scan_block(dxdy[a:b].contiguous(), matched[a:b].contiguous(), first_is_start=(r == 0))
                     for r, (a, b) in enumerate(zip(cuts[:-1], cuts[1:]))]
            summ = torch.cat([p[2] for p in parts]).contiguous()
            for r, (pp, ff, _) in enumerate(parts):
                ctx.frc_apply_blocks(summ, r, world, pp, ff)
            nfrag = ctx.frc_finish()
            assert nfrag == carry[1] + 1
            assert torch.equal(torch.cat([p[0] for p in parts]), pos), (trial, world)
            assert torch.equal(torch.cat([p[1] for p in parts]), frag), (trial, world)
    ctx.close()


def _boxnoise(seed, shift=0, box=(6, 6, 86, 126), base=None):
    """a frame whose box is pure noise: ~6 000 keypoints in grid region 0 (and hundreds in its
    neighbours), more than the 4 096 entries the LDS instances of the matcher hold per region"""
    rng = np.random.default_rng(seed)
    img = np.full((240, 320), 6, np.uint8) if base is None else base.copy()
    x0, y0, x1, y1 = box
    img[y0:y1, x0:x1] = rng.integers(0, 16, (y1 - y0, x1 - x0))
    return np.roll(img, -shift, axis=1)


def test_grid_matcher_regions_beyond_the_lds_instances(gpu):
    """Frames with more than 4 096 entries in a grid region, alone and in matching pairs: the LDS
    instances leave their tickets to the global-memory matcher (rmx_kpm_status flag 128,
    rmx_kpm_resolve); tickets, offsets and positions equal the oracle's."""
    import torch

    frames, _ = scroller_v1(12, dens=100)
    frames = frames.copy()
    frames[4] = _boxnoise(7)
    frames[5] = _boxnoise(7, shift=2)   # a matching pair of huge regions
    frames[6] = _boxnoise(7, shift=5)
    frames[9] = _boxnoise(8, base=frames[9])  # a huge region in an ordinary frame
    for i in (4, 9):
        _, dsc, k = O.orc_kpe(frames[i])
        assert k <= 16384 and int((dsc["region_mask"] & 1).sum()) > 4096
    n, H, W = frames.shape
    ctx = L.Context(W, H, kmax=16384)
    d = torch.from_numpy(frames).cuda()
    med, desc, kc = ctx.kpe(d)
    assert ctx.kpe_status()[0] == 0
    dxdy, matched = ctx.kpm_seq(desc, kc)
    rc, fl = ctx.kpm_status()
    assert rc == L.RMX_ERR_CAPACITY and fl == 128, (rc, fl)
    nres = ctx.kpm_resolve(desc, kc, dxdy, matched)
    assert nres > 0
    tickets = ctx.kpm_tickets(n)
    pos, frag, carry = ctx.frc_scan(dxdy, matched)
    o = O.orc_frc(frames)
    assert np.array_equal(dxdy.cpu().numpy(), o["dxdy"]) and np.array_equal(matched.cpu().numpy(), o["matched"])
    assert np.array_equal(pos.cpu().numpy(), o["pos"]) and np.array_equal(frag.cpu().numpy(), o["frag"])
    tabs = L.desc_to_numpy(desc, kc)
    for q in (4, 5, 6, 7, 9, 10):
        m, dd, t = O.orc_kpm_grid(tabs[q - 1], tabs[q])
        for r in range(8):
            for j in range(3):
                a, b = tickets[q, r, j], t[r, j]
                assert int(a["valid"]) == int(b["valid"]), (q, r, j)
                if int(b["valid"]):
                    assert (int(a["dx"]), int(a["dy"]), int(a["count"])) == (int(b["dx"]), int(b["dy"]), int(b["count"])), (q, r, j)
    # a second call finds nothing left
    assert ctx.kpm_resolve(desc, kc, dxdy, matched) == 0
    ctx.close()


def test_grid_matcher_frames_beyond_16384_keypoints(gpu):
    """Pure-noise frames: ~46 000 keypoints each, more than the 16 384 rows the matcher's pool is
    laid out for — such a frame is left empty there and all eight tickets of its two pairs are
    computed by the global-memory instance (tables of kmax = 65 536 rows from kpe_table_big)."""
    import torch

    frames, _ = scroller_v1(8, dens=100)
    frames = frames.copy()
    rng = np.random.default_rng(12)
    noise = rng.integers(0, 16, size=(240, 320), dtype=np.uint8)
    frames[3] = noise
    frames[4] = np.roll(noise, -2, axis=1)
    frames[5] = np.roll(noise, -4, axis=1)
    assert O.orc_kpe(noise)[2] > 16384
    n, H, W = frames.shape
    ctx = L.Context(W, H, kmax=65536)
    d = torch.from_numpy(frames).cuda()
    med, desc, kc = ctx.kpe(d)
    assert ctx.kpe_status()[0] == 0
    dxdy, matched = ctx.kpm_seq(desc, kc)
    rc, fl = ctx.kpm_status()
    assert rc == L.RMX_ERR_CAPACITY and fl == 128, (rc, fl)
    assert ctx.kpm_resolve(desc, kc, dxdy, matched) >= 32
    pos, frag, carry = ctx.frc_scan(dxdy, matched)
    o = O.orc_frc(frames)
    assert np.array_equal(dxdy.cpu().numpy(), o["dxdy"]) and np.array_equal(matched.cpu().numpy(), o["matched"])
    assert np.array_equal(pos.cpu().numpy(), o["pos"]) and np.array_equal(frag.cpu().numpy(), o["frag"])
    assert matched.cpu().numpy()[4] == 1 and tuple(dxdy.cpu().numpy()[4]) == (2, 0)
    tickets = ctx.kpm_tickets(n)
    tabs = L.desc_to_numpy(desc, kc)
    for q in (3, 4, 6):
        m, dd, t = O.orc_kpm_grid(tabs[q - 1], tabs[q])
        for r in range(8):
            for j in range(3):
                a, b = tickets[q, r, j], t[r, j]
                assert int(a["valid"]) == int(b["valid"]), (q, r, j)
                if int(b["valid"]):
                    assert (int(a["dx"]), int(a["dy"]), int(a["count"])) == (int(b["dx"]), int(b["dy"]), int(b["count"])), (q, r, j)
    ctx.close()


def test_aligner_resolves_huge_regions_by_itself(gpu):
    import torch

    from remap_amd.pipeline import Aligner

    frames, _ = scroller_v1(10, dens=100)
    frames = frames.copy()
    frames[3] = _boxnoise(17)
    frames[4] = _boxnoise(17, shift=3)
    al = Aligner(320, 240, kmax=16384)
    out = al.align(torch.from_numpy(frames).cuda())
    o = O.orc_frc(frames)
    assert np.array_equal(out["dxdy"].cpu().numpy(), o["dxdy"]) and np.array_equal(out["pos"].cpu().numpy(), o["pos"])
    assert out["nfrag"] == int(o["frag"].max()) + 1
    al.close()


def test_deferred_steps_resolve_huge_regions(gpu):
    """The deferred (pipelined) path settles every step before its lane is used again: a block whose
    tickets were left to rmx_kpm_resolve is resolved and scanned again, exactly like the synchronous
    path — on one Aligner with steps enqueued back to back, and through AlignerPipe (the bench's
    headline path), where a later block on the same lane would otherwise reset the work list."""
    import torch

    from remap_amd.pipeline import Aligner, AlignerPipe

    frames, _ = scroller_v1(10, dens=100)
    frames = frames.copy()
    frames[3] = _boxnoise(17)
    frames[4] = _boxnoise(17, shift=3)
    plain, _ = scroller_v1(10, dens=100)
    o_big, o_plain = O.orc_frc(frames), O.orc_frc(plain)
    d_big, d_plain = torch.from_numpy(frames).cuda(), torch.from_numpy(plain).cuda()

    def same(out, o):
        return (np.array_equal(out["dxdy"].cpu().numpy(), o["dxdy"]) and np.array_equal(out["pos"].cpu().numpy(), o["pos"])
                and np.array_equal(out["frag"].cpu().numpy(), o["frag"]) and out["nfrag"] == int(o["frag"].max()) + 1)

    al = Aligner(320, 240, kmax=16384)
    al.align(d_plain, defer=True)
    out = al.align(d_big, defer=True)
    snap = {k: (v.clone() if hasattr(v, "clone") else v) for k, v in out.items()}  # (settled when the next step starts)
    al.align(d_plain, defer=True)
    snap.update(pos=out["pos"].clone(), frag=out["frag"].clone(), dxdy=out["dxdy"].clone(), nfrag=out["nfrag"])
    assert same(snap, o_big)
    last = al.align(d_big, defer=True)
    assert al.finish() == int(o_big["frag"].max()) + 1 and same(last, o_big)
    al.close()

    pipe = AlignerPipe(320, 240, kmax=16384, lanes=2)
    outs = []
    for blk in (d_plain, d_big, d_big, d_plain, d_big):
        out = pipe.align(blk)
        outs.append((out, o_big if blk is d_big else o_plain))
    # lanes alternate: blocks 0, 2, 4 on lane 0 and 1, 3 on lane 1; a lane's buffers hold its LAST block
    assert pipe.finish() == int(o_big["frag"].max()) + 1
    assert same(outs[4][0], outs[4][1]) and same(outs[3][0], outs[3][1])
    # through submit / drain every block is returned complete
    pipe.submit(d_big)
    done = [pipe.submit(d_plain), pipe.submit(d_big)] + pipe.drain()
    done = [d for d in done if d is not None]
    assert len(done) == 3 and same(done[0], o_big) and same(done[1], o_plain) and same(done[2], o_big)
    pipe.close()


@pytest.mark.parametrize("n,kind", [(1, "random"), (13, "random"), (14, "random"), (5000, "random"), (5088, "few"), (120000, "random"),
                                    (90000, "few"), (400000, "random"), (30000, "equal")])
def test_unordered_map_order_replay_at_scale(gpu, n, kind):
    """umap_order_big (the tie-breaking order of the global-memory matchers) against a real
    std::unordered_map (oracle, libstdc++): random hashes, hashes that fall into a handful of
    buckets, all-equal hashes; sizes across the rehash points 13, 29, ... 351 061."""
    import torch

    rng = np.random.default_rng(n)
    if kind == "random":
        h = rng.integers(0, 2**63, size=n, dtype=np.uint64) * 2 + rng.integers(0, 2, size=n, dtype=np.uint64)
    elif kind == "few":
        h = rng.integers(0, 7, size=n, dtype=np.uint64) * np.uint64(1000003)
    else:
        h = np.full(n, 12345678901234567, np.uint64)
    want = O.orc_umap_order_std(h)
    ctx = L.Context(16, 8, grid=(1, 1), overlap=0, kmax=64)
    got = ctx.debug_umap_order(torch.from_numpy(h.view(np.int64)).cuda()).cpu().numpy().astype(np.uint32)
    ctx.close()
    assert np.array_equal(got, want)


def test_adaptive_launch_set_learns_and_recovers(gpu):
    """rmx_kpm_adaptive (what Aligner switches on): the first step launches every instance of the matcher's
    chains and keeps those that found work; a later block that needs another one (a dense block after sparse
    ones: wave-per-region builds, the per-(pair, region) match tiers) is flagged, redone with all of them by
    the resolve pass of the step's check, and the set grows — offsets, positions and fragment ids equal those
    of a context that always launches everything, block for block."""
    import torch

    from remap_amd.pipeline import Aligner

    sparse, _ = scroller_v1(300, dens=200)
    dense, _ = scroller_v1(300, dens=40, cam_seed=5)
    d_sparse, d_dense = torch.from_numpy(sparse).cuda(), torch.from_numpy(dense).cuda()
    ref = Aligner(320, 240, kmax=4096)
    ref.ctx.kpm_adaptive(False)
    assert ref.ctx.kpm_launch_set() == 127
    want = {}
    for name, blk in (("sparse", d_sparse), ("dense", d_dense)):
        o = ref.align(blk)
        want[name] = ({k: o[k].clone() for k in ("dxdy", "matched", "pos", "frag")}, o["nfrag"])
    ref.close()

    def same(out, name):
        w, nf = want[name]
        return out["nfrag"] == nf and all(torch.equal(out[k], v) for k, v in w.items())

    al = Aligner(320, 240, kmax=4096)
    assert al.ctx.kpm_launch_set() == 127  # nothing learned yet: the first step launches all
    assert same(al.align(d_sparse), "sparse")
    learned = al.ctx.kpm_launch_set()
    assert learned != 127 and not learned & 64  # (no sparse frame reaches the 136-KB match tier)
    assert same(al.align(d_sparse), "sparse") and al.ctx.kpm_launch_set() == learned
    assert same(al.align(d_dense), "dense")  # needs instances outside the set: redone, same results
    grown = al.ctx.kpm_launch_set()
    assert grown & learned == learned and grown != learned and grown & 2  # (kpm_build_medium<256> builds dense frames)
    assert same(al.align(d_dense), "dense") and al.ctx.kpm_launch_set() == grown
    # deferred steps: the flagged step is settled (and redone) before its buffers are used again
    al2 = Aligner(320, 240, kmax=4096)
    al2.align(d_sparse, defer=True)
    al2.align(d_sparse, defer=True)
    out = al2.align(d_dense, defer=True)
    al2.finish()
    assert same(out, "dense")
    al2.close()
    al.close()


def test_redone_step_that_hits_a_hard_limit_is_not_returned_as_good(gpu):
    """The adaptive launch set: a step that needed an instance outside the set is redone by rmx_kpm_resolve with
    every instance — the ONLY pass that builds the frames the first one skipped.  If that pass raises a matcher
    limit other than "tickets left" (status word 1, sticky) and leaves no ticket for the global-memory instance,
    resolve used to return RMX_OK from its "nothing left to do" exits and the step's offsets went out as good.
    The limit is injected (rmx_debug_poke_status) between the first pass and the redo."""
    import torch

    from remap_amd.pipeline import Aligner

    sparse, _ = scroller_v1(200, dens=200)
    dense, _ = scroller_v1(200, dens=40, cam_seed=5)
    d_sparse, d_dense = torch.from_numpy(sparse).cuda(), torch.from_numpy(dense).cuda()
    ctx = L.Context(320, 240, kmax=4096)
    ctx.kpm_adaptive(True)
    med, desc, kc = ctx.kpe(d_sparse)
    dxdy, matched = ctx.kpm_seq(desc, kc)
    assert ctx.kpm_status() == (0, 0) and ctx.kpm_launch_set() != 127  # learned: the sparse stream's instances only
    ctx.kpe(d_dense, med, desc, kc)
    dxdy, matched = ctx.kpm_seq(desc, kc)
    rc, fl = ctx.kpm_status()
    assert rc == L.RMX_ERR_CAPACITY and fl == 128  # "redo me": frames waiting for an instance that was not launched
    ctx._chk(ctx.lib.rmx_debug_poke_status(ctx.h, 1, 4), "poke")  # ... and the redo "hits" a hard limit
    with pytest.raises(L.RemapError) as ei:
        ctx.kpm_resolve(desc, kc, dxdy, matched)
    assert ei.value.code == L.RMX_ERR_CAPACITY
    assert ctx.kpm_status() == (0, 0)  # reported once
    # the same through the Aligner: the error surfaces at the step it belongs to, the Aligner goes on
    al = Aligner(320, 240, kmax=4096)
    al.align(d_sparse)
    out = al.align(d_dense, defer=True)
    torch.cuda.synchronize()
    al.ctx.lib.rmx_debug_poke_status(al.ctx.h, 1, 4)
    with pytest.raises(L.RemapError):
        al.finish()
    ok = al.align(d_dense)
    assert ok["nfrag"] >= 1
    al.close()
    ctx.close()
