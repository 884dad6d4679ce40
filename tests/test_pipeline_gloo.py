"""Multi-rank host logic on CPU: world_size-2 gloo run of the sharding + the two
all-gathers of remap_amd/pipeline.py (boundary descriptor tables, per-frame offsets).
The per-rank compute stand-in is the CPU oracle (allowed in tests); the result must equal
the single-process alignment bit for bit."""
import os
import socket

import numpy as np
import pytest

import orclib as O
from remap_amd.pipeline import gather_boundary_tables, gather_offsets, shard_range
from remap_amd.synth import scroller_v1

KMAX = 512
N_TOTAL = 41  # uneven blocks: 21 + 20
W, H = 160, 64


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _stream():
    fr, truth = scroller_v1(N_TOTAL, width=W, height=H, dens=50)
    fr = fr.copy()
    fr[21] = 6  # a break right at the rank boundary's second frame
    return fr, truth


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    fr, _ = _stream()
    lo, hi = shard_range(N_TOTAL, rank, world)
    n = hi - lo
    desc = np.zeros((n, KMAX, 18), np.uint8)
    kc = np.zeros(n, np.int32)
    tabs = []
    for i in range(n):
        _, d, k = O.orc_kpe(fr[lo + i])
        assert k <= KMAX
        desc[i, :k] = d.view(np.uint8).reshape(-1, 18)
        kc[i] = k
        tabs.append(d)
    tdesc, tk = torch.from_numpy(desc), torch.from_numpy(kc)
    btabs, bcnt = gather_boundary_tables(tdesc[n - 1], tk[n - 1:n], dist)
    assert btabs.shape == (world, KMAX, 18) and bcnt.shape == (world,)
    dxdy = np.zeros((n, 2), np.int32)
    matched = np.zeros(n, np.uint8)
    for i in range(n):
        if i == 0:
            if rank == 0:
                continue
            k = int(bcnt[rank - 1])
            prev = btabs[rank - 1].numpy()[:k].reshape(-1).view(O.DESC_DTYPE)
        else:
            prev = tabs[i - 1]
        m, d, _ = O.orc_kpm_grid(prev, tabs[i])
        matched[i] = m
        dxdy[i] = d
    counts = [shard_range(N_TOTAL, r, world)[1] - shard_range(N_TOTAL, r, world)[0] for r in range(world)]
    gd, gm = gather_offsets(torch.from_numpy(dxdy), torch.from_numpy(matched), counts, dist)
    q.put((rank, gd.numpy().copy(), gm.numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_range_covers_stream():
    for n in (1, 7, 41, 10000):
        for w in (1, 2, 3, 8):
            r = [shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[i][1] == r[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in r]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(300)
def test_two_rank_gloo_equals_single_process():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    fr, truth = _stream()
    o = O.orc_frc(fr)
    for rank, gd, gm in res:
        assert np.array_equal(gd, o["dxdy"]), rank
        assert np.array_equal(gm, o["matched"]), rank
    assert o["matched"][21] == 0 and o["matched"][22] == 0 and o["matched"][20] == 1


def _scan(dxdy, matched, first_is_start, carry=((0, 0), 0)):
    """frc.hpp:109-115 on the host: positions, fragment ids, carry-out."""
    (x, y), c = carry
    pos, frag = [], []
    for i in range(len(matched)):
        if (i == 0 and first_is_start) or not matched[i]:
            c += 0 if (i == 0 and first_is_start) else 1
            x = y = 0
        else:
            x += int(dxdy[i][0])
            y += int(dxdy[i][1])
        pos.append((x, y))
        frag.append(c)
    return np.array(pos).reshape(-1, 2), np.array(frag), ((x, y), c)


def test_block_summaries_stitch_to_the_global_scan():
    """fold_block_summaries: per-rank scans + 16-byte summaries reproduce the scan of the whole
    stream (any number of ranks, blocks without any fragment start, starts at block edges)."""
    from remap_amd.pipeline import fold_block_summaries

    rng = np.random.default_rng(12)
    for trial in range(40):
        n, world = int(rng.integers(5, 200)), int(rng.integers(1, 9))
        p_break = float(rng.choice([0.0, 0.02, 0.3]))
        matched = (rng.random(n) >= p_break).astype(np.uint8)
        dxdy = rng.integers(-4, 5, size=(n, 2))
        gpos, gfrag, gcarry = _scan(dxdy, matched, True)
        summ = []
        for r in range(world):
            lo, hi = shard_range(n, r, world)
            _, _, ((x, y), c) = _scan(dxdy[lo:hi], matched[lo:hi], r == 0)
            # bit 1 of the first word: "this block has tickets left to rmx_kpm_resolve" — carried
            # along by the exchange, ignored by the fold
            summ.append([int(c > 0 or r == 0) | (2 if rng.random() < 0.3 else 0), x, y, c])
        assert sum(s[3] for s in summ) == gcarry[1]
        for r in range(world):
            lo, hi = shard_range(n, r, world)
            if hi == lo:
                continue
            carry = fold_block_summaries(summ, r)
            pos, frag, _ = _scan(dxdy[lo:hi], matched[lo:hi], r == 0, carry if r else ((0, 0), 0))
            assert np.array_equal(pos, gpos[lo:hi]) and np.array_equal(frag, gfrag[lo:hi]), (trial, r)
