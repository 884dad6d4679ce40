"""Two ranks on ONE GPU (gloo for the exchanges, both processes on cuda:0): the frame-sharded
collect (boundary-table and offset all-gathers), the cross-rank canvas sum and the pair-sharded
fgs::splice must reproduce the reference's golden collect -> splice result on every rank."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch
    import torch.distributed as dist

    from remap_amd.pipeline import Aligner, shard_range
    from remap_amd.splice import splice
    from remap_amd.synth import rooms_v1

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = np.load(os.path.join(G, "fgs_splice.npz"))
    n, w, h, rooms, dens, seed = g["gen"].tolist()
    frames, _, _ = rooms_v1(n, width=w, height=h, n_rooms=rooms, dens=dens, seed=seed)
    lo, hi = shard_range(n, rank, world)
    local = torch.from_numpy(np.ascontiguousarray(frames[lo:hi])).cuda()
    al = Aligner(w, h, kmax=2048, dist=dist)
    out = al.align(local, n_total=n, blit=True)
    frags = al.fragments(local, out)
    ncollected = len(frags)
    res = splice(frags, dist=dist)
    from test_oracle_golden import poly_sum

    summary = [(list(f.dims), poly_sum(f.canvas.cpu().numpy().view(np.uint16)), sorted(f.frames)) for f in res]
    q.put((rank, ncollected, summary))
    al.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
def test_two_ranks_collect_sum_splice_equal_reference(gpu):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=500) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    g = np.load(os.path.join(G, "fgs_splice.npz"))
    for rank, ncollected, summary in res:
        assert ncollected == int(g["ncollected"]), rank
        assert len(summary) == int(g["nfrag"]), rank
        for k, (dims, csum, fr) in enumerate(summary):
            assert dims == g["dims"][k].tolist(), (rank, k)
            assert csum == int(g["canvas_sum"][k]), (rank, k)
            for no, p in fr:
                assert int(g["frag"][no]) == k and list(p) == g["pos"][no].tolist(), (rank, k, no)


def _worker_align(rank, world, port, q):
    import torch
    import torch.distributed as dist

    from remap_amd.pipeline import Aligner, shard_range
    from remap_amd.synth import scroller_v1

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    frames = _broken_stream()
    n = len(frames)
    lo, hi = shard_range(n, rank, world)
    al = Aligner(320, 240, kmax=2048, dist=dist)
    out = al.align(torch.from_numpy(np.ascontiguousarray(frames[lo:hi])).cuda(), n_total=n, blit=True)
    res = (rank, lo, hi, out["dxdy"].cpu().numpy(), out["matched"].cpu().numpy(), out["pos"].cpu().numpy(),
           out["frag"].cpu().numpy(), out["pos_all"].cpu().numpy(), out["frag_all"].cpu().numpy(), out["nfrag"])
    al.close()
    # the same block through two lanes in flight (AlignerPipe shares the process group): three
    # steps, same offsets / positions / fragment ids / fragment count as the one above
    from remap_amd.pipeline import AlignerPipe

    pipe = AlignerPipe(320, 240, kmax=2048, dist=dist, lanes=2)
    local = torch.from_numpy(np.ascontiguousarray(frames[lo:hi])).cuda()
    outs = [pipe.align(local, n_total=n) for _ in range(3)]
    nf = pipe.finish()
    same = nf == res[9]
    for o in outs[1:]:
        same = same and all(np.array_equal(o[k].cpu().numpy(), v) for k, v in (("dxdy", res[3]), ("matched", res[4]), ("pos", res[5]), ("frag", res[6])))
    pipe.close()
    q.put(res + (bool(same),))
    dist.barrier()
    dist.destroy_process_group()


def _broken_stream():
    from remap_amd.synth import scroller_v1

    frames, _ = scroller_v1(203, dens=120)
    frames = frames.copy()
    for i in (50, 51, 101, 150):  # breaks inside blocks and exactly at a block edge of the 4-rank cut
        frames[i] = 6
    return frames


@pytest.mark.timeout(600)
def test_four_ranks_equal_single_process_bit_for_bit(gpu):
    """Config 3 in miniature: the frame-sharded alignment (4 ranks, uneven blocks, fragment
    breaks at and inside block edges) gives every rank exactly the single-process offsets,
    positions and fragment ids for its frames, and the same whole-stream view."""
    import torch
    import torch.multiprocessing as mp

    from remap_amd.pipeline import Aligner

    frames = _broken_stream()
    al = Aligner(320, 240, kmax=2048)
    ref = al.align(torch.from_numpy(frames).cuda())
    want = {k: ref[k].cpu().numpy() for k in ("dxdy", "matched", "pos", "frag")}
    nfrag = ref["nfrag"]
    al.close()
    assert nfrag >= 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_align, args=(r, 4, port, q)) for r in range(4)]
    for p in procs:
        p.start()
    res = [q.get(timeout=500) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, lo, hi, dxdy, matched, pos, frag, pos_all, frag_all, nf, pipe_same in res:
        first = 1 if rank == 0 else 0
        assert pipe_same, rank
        assert np.array_equal(matched, want["matched"][lo:hi]), rank
        assert np.array_equal(dxdy[first:], want["dxdy"][lo + first:hi]), rank
        assert np.array_equal(pos, want["pos"][lo:hi]) and np.array_equal(frag, want["frag"][lo:hi]), rank
        assert np.array_equal(pos_all, want["pos"]) and np.array_equal(frag_all, want["frag"]), rank
        assert nf == nfrag


def _huge_region_stream():
    from remap_amd.synth import scroller_v1

    frames, _ = scroller_v1(40, dens=120)
    frames = frames.copy()
    rng = np.random.default_rng(4)
    big = np.full((240, 320), 6, np.uint8)
    big[6:126, 6:86] = rng.integers(0, 16, (120, 80))  # ~6 000 keypoints in grid region 0
    frames[31] = big
    frames[32] = np.roll(big, -3, axis=1)
    return frames


def _worker_resolve(rank, world, port, q):
    import torch
    import torch.distributed as dist

    from remap_amd.pipeline import Aligner, shard_range

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    frames = _huge_region_stream()
    n = len(frames)
    lo, hi = shard_range(n, rank, world)
    local = torch.from_numpy(np.ascontiguousarray(frames[lo:hi])).cuda()
    if os.environ.get("RMX_TEST_PIPE"):  # the pipelined path: two lanes sharing the process group
        from remap_amd.pipeline import AlignerPipe

        al = AlignerPipe(320, 240, kmax=16384, dist=dist, lanes=2)
        al.align(local, n_total=n)
        al.align(local, n_total=n)
        out = al.align(local, n_total=n)
        out["nfrag"] = al.finish()
    else:
        al = Aligner(320, 240, kmax=16384, dist=dist)
        out = al.align(local, n_total=n)
    q.put((rank, lo, hi, out["dxdy"].cpu().numpy(), out["matched"].cpu().numpy(), out["pos"].cpu().numpy(),
           out["frag"].cpu().numpy(), out["nfrag"]))
    al.close()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("pipelined", [False, True])
def test_two_ranks_resolve_a_huge_region_together(gpu, pipelined, monkeypatch):
    """Only rank 1's block holds frames with more than 4 096 keypoints in one grid region.  The
    flag travels with the scan summaries: both ranks take the resolve-and-scan-again path (a
    no-op on rank 0), their collectives pair up, and the result equals the oracle's — also when the
    blocks go through AlignerPipe (deferred steps, settled before a lane is used again)."""
    import torch.multiprocessing as mp

    if pipelined:
        monkeypatch.setenv("RMX_TEST_PIPE", "1")

    import orclib as O

    frames = _huge_region_stream()
    o = O.orc_frc(frames)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_resolve, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=500) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, lo, hi, dxdy, matched, pos, frag, nf in res:
        assert np.array_equal(matched, o["matched"][lo:hi]) and np.array_equal(dxdy, o["dxdy"][lo:hi]), rank
        assert np.array_equal(pos, o["pos"][lo:hi]) and np.array_equal(frag, o["frag"][lo:hi]), rank
        assert nf == int(o["frag"].max()) + 1


@pytest.mark.timeout(600)
def test_bench_two_ranks_under_torch_distributed_run(gpu):
    """bench.py exactly as the driver launches it for N > 1 (`python -m torch.distributed.run --nproc-per-node N
    ... bench.py --gpus N`), two ranks on this one card with gloo for the exchanges: the per-rank block from
    scroller_world, the lanes' process groups, the cross-rank max of the timed region and the exactness check —
    so that an 8-GPU run is not the first execution of any of those lines.  Launched from a fresh process."""
    import json
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--no-cpu",
           "--no-workloads", "--headline-only", "--steps", "3", "--warmup", "1", "--spin-up", "0", "--frames", "2000"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=540)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # rank 0 prints ONE line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak"
    assert d["offsets_bit_exact_vs_truth"] is True
    assert d["config"]["frames_total"] == 4000 and "x2" in d["config"]["parallelism"]
    assert d["value"] > 0 and abs(d["value"] - 4000 / (d["ms_per_step"] * 1e-3)) < 1e-3 * d["value"]
