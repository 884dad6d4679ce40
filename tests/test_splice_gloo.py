"""Multi-rank fgs host logic on CPU (SURVEY.md §8e, rows 'fgm blit' and 'fgs all-pairs'):
world_size-2 gloo runs of
  * score_pairs_sharded — the block-cyclic cut of the fragment pair list + ONE all-gather of
    the vote records; the per-rank compute stand-in is the CPU oracle (allowed in tests) on
    the reference's golden cellular-match cases, the gathered vote vector must equal the
    golden results pair for pair on every rank;
  * sum_subcanvases — per-rank sub-canvases of one fragment summed to the reference canvas,
    with uint16 wraparound across ranks."""
import os
import socket

import numpy as np
import pytest

import orclib as O
from remap_amd import lib as L
from remap_amd.pipeline import sum_subcanvases
from remap_amd.splice import score_pairs_sharded, shard_pairs

G = os.path.join(os.path.dirname(__file__), "golden")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _desc_of(a):
    return np.ascontiguousarray(a).reshape(-1).view(O.DESC_DTYPE)


def _canvas_case():
    """One fragment, two ranks: rank 0 covers steps [0, 2), rank 1 steps [1, 3) of a 3-step
    wide box; bins near 65535 on both so the overlap wraps."""
    rng = np.random.default_rng(3)
    subs = []
    for r in range(2):
        c = rng.integers(0, 4, size=(8, 32, 16)).astype(np.uint16)
        c[2:5, (slice(10, 20), slice(0, 4))[r], 3] = 65000 + r * 400  # global columns 16..19: 65000 + 65400 wraps
        subs.append(c)
    zeros = [(-16, 0), (0, 0)]       # canvas coordinates of each sub-box's top-left
    gbox = (48, 8, -16, 0)           # whole box: 48 wide, zero at x = -16
    want = np.zeros((8, 48, 16), np.uint16)
    want[:, 0:32] += subs[0]
    want[:, 16:48] += subs[1]        # numpy uint16 addition wraps like fgm::dot_type
    return subs, zeros, gbox, want


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = np.load(os.path.join(G, "kpm_cell_cases.npz"))
    npairs = int(g["npairs"])
    scored = []

    def score(idx):
        out = np.zeros(len(idx), L.VOTE_DTYPE)
        for o, k in enumerate(idx):
            _, _, _, _, i, j = g[f"res{k}"].tolist()
            m, v = O.orc_kpm_cell(_desc_of(g[f"desc{i}"]), g[f"mask{i}"], _desc_of(g[f"desc{j}"]), g[f"mask{j}"].shape)
            out[o] = (v["dx"], v["dy"], v["count"], 1) if m else (0, 0, 0, 0)
            scored.append(k)
        return out

    votes = score_pairs_sharded(score, npairs, dist)
    subs, zeros, gbox, _ = _canvas_case()
    whole = sum_subcanvases(torch.from_numpy(subs[rank].view(np.int16)), zeros[rank], gbox, dist)
    # a fragment without frames on this rank contributes nothing
    only0 = sum_subcanvases(torch.from_numpy(subs[0].view(np.int16)) if rank == 0 else None, zeros[0], gbox, dist)
    q.put((rank, votes.copy(), sorted(scored), whole.numpy().view(np.uint16).copy(), only0.numpy().view(np.uint16).copy()))
    dist.barrier()
    dist.destroy_process_group()


def test_shard_pairs_is_a_partition():
    for n in (0, 1, 5, 12, 101):
        for w in (1, 2, 3, 8):
            parts = [shard_pairs(n, r, w) for r in range(w)]
            assert sorted(sum(parts, [])) == list(range(n))
            assert max(map(len, parts)) - min(map(len, parts)) <= 1


@pytest.mark.timeout(300)
def test_two_rank_gloo_pair_votes_and_canvas_sum():
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
    g = np.load(os.path.join(G, "kpm_cell_cases.npz"))
    npairs = int(g["npairs"])
    subs, zeros, gbox, want = _canvas_case()
    only0_want = np.zeros_like(want)
    only0_want[:, 0:32] = subs[0]
    for rank, votes, scored, whole, only0 in res:
        assert scored == shard_pairs(npairs, rank, 2)  # every rank scored only its own share
        for k in range(npairs):
            m, dx, dy, cnt, _, _ = g[f"res{k}"].tolist()
            assert int(votes[k]["valid"]) == m, (rank, k)
            if m:
                assert (int(votes[k]["dx"]), int(votes[k]["dy"]), int(votes[k]["count"])) == (dx, dy, cnt), (rank, k)
        assert np.array_equal(whole, want), rank
        assert int(whole[3, 17, 3]) == (65000 + 65400) % 65536
        assert np.array_equal(only0, only0_want), rank
