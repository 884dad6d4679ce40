#!/usr/bin/env python3
"""bench.py — aligned frames/s of the MI355X hot path on BASELINE.json's workload.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A step = one pass of the hot path (kpe -> descriptor tables -> kpm -> frc position scan)
over this rank's block of synthetic frames, already resident in HBM; the K timed steps are
enqueued back to back and the host waits once at the end of the timed region (with the status
words of all K steps checked there).  At N = 1 the workload
is BASELINE.json configs[1]: 10 000 scroller-v1 frames of 320x240; at N > 1 every rank gets
its own 10 000-frame block of ONE N*10 000-frame stream (weak scaling), boundary descriptor
tables and per-frame offsets are all-gathered over RCCL, and the result is the global
alignment.  Rank 0 prints ONE JSON line.

The line also carries
  roofline     — the extraction kernel (the dominant one: candidate test on bit planes and 5x5 finish in one
                 launch; kpe_pc on this stream, `roofline.kernel` names the one that ran): algorithmic
                 bytes (2*W*H + 18*K per frame) / its average launch time from HIP events
                 on the launch stream, against the 8 TB/s HBM peak; `one_block_in_flight` = the same
                 kernel with the chip to itself; `previous_kernels` = kpe_mf (round 2) and the two-kernel
                 form of rounds 1-2a, timed in this run on this box.  Every figure is measured in the run, except
                 `traffic` and `valu_wave_instructions`, which come from the committed rocprofv3 --pmc
                 passes named in `*_source` (counters cannot be read from inside the process);
  sustained    — the timed region repeated for a second (K-step groups: median / min / max), so the headline does
                 not rest on one 8-ms window;
  workloads    — the other BASELINE.json workloads, timed in this run (value + the extraction kernel's
                 roofline fraction each, `profiles`: the committed rocprofv3 summaries of that workload): dense
                 stream, 640x480 native and pixel-doubled, rooms (collect with canvas blit, then fgs::splice);
  cpu_baseline — the reference's own frc::collector::collect (oracle/_ref, built from
                 /root/reference by oracle/build_ref.sh) or, if that library is absent, the
                 CPU restatement (oracle/), timed on one host core on the same frames; `all_cores` =
                 SURVEY.md §8d (ii): frame-parallel kpe on every host core + sequential kpm/frc.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

W, H = 320, 240
FRAMES_PER_GPU = 10000
DENS = 200
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)


KERNEL_NOTES = {
    "kpe_pc": "candidate test by pairwise rank comparison on bit planes, 32 pixels per lane operation; 5x5 finish rolled into the row "
              "loop, windows from an LDS ring of the streamed rows; the median plane = the input rows from that ring with the "
              "candidates' nibbles patched",
    "kpe_pc_dense": "kpe_pc with the median patches of eight passes per roll in registers (dense streams)",
    "kpe_bs_ring": "3x3 rank selection on bit planes, 32 pixels per lane operation, + 5x5 finish rolled into the row loop, windows "
                   "from an LDS ring of the streamed rows",
    "kpe_bs": "3x3 rank selection on bit planes, 32 pixels per lane operation, + 5x5 finish of the workgroup's own candidates",
}


def cpu_baseline(frames_host):
    """Reference (or port) CPU path on one host core over `frames_host`.

    The reference's collect cannot skip its blit, so `value` covers kpe+kpm+frc+fgm blit — the
    stages of the GPU's `frames_per_s_with_blit`, not of the headline; `alignment_only` times the
    CPU restatement's kpe+kpm+frc loop (the headline's stages) on a shorter sample."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import orclib as O  # oracle/: allowed here only as the reported CPU baseline

    n = len(frames_host)
    def timed(fn, frames, budget):  # passes over the sample until ~budget s of CPU work are on the clock (at most 4)
        t0, passes = time.perf_counter(), 0
        while passes < 4 and (passes == 0 or time.perf_counter() - t0 < budget):
            fn(frames)
            passes += 1
        return time.perf_counter() - t0, passes

    m = min(n, 4000)
    dta, pa = timed(O.orc_frc, frames_host[:m], 4.0)
    port = {"value": m * pa / dta, "unit": "frames/s", "cores": 1, "kind": "port", "stages": "kpe+kpm+frc",
            "sample": f"oracle/remap_oracle.cpp kpe+kpm+frc loop (scalar C++, no blit) on the first {m} frames of the "
                      f"same stream, {pa} passes, {dta:.1f} s"}
    # kpe of the frames on all host cores (one oracle call per frame, the GIL is released
    # inside), then the sequential part: kpm between consecutive tables + the position scan
    from concurrent.futures import ThreadPoolExecutor

    cores = os.cpu_count() or 1
    mc = min(n, 250 * cores)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(cores) as ex:
        tabs = list(ex.map(lambda f: O.orc_kpe(f)[1], frames_host[:mc]))
    t_kpe = time.perf_counter() - t0
    t0 = time.perf_counter()
    for i in range(1, mc):
        O.orc_kpm_grid(tabs[i - 1], tabs[i])
    t_kpm = time.perf_counter() - t0
    port["all_cores"] = {"value": mc / (t_kpe + t_kpm), "unit": "frames/s", "cores": cores, "kind": "port",
                         "stages": "kpe (frame-parallel) + kpm/frc (sequential)",
                         "sample": f"first {mc} frames: kpe {t_kpe:.2f} s on {cores} threads + kpm {t_kpm:.2f} s on one"}
    if O.ref_full() is not None:
        dt, passes = timed(O.ref_frc, frames_host, 10.0)
        return {"value": n * passes / dt, "unit": "frames/s", "cores": 1, "kind": "reference",
                "stages": "kpe+kpm+frc+fgm blit (compare with frames_per_s_with_blit)",
                "sample": f"reference frc::collector::collect (kpe+kpm+fgm blit, no-op codec, g++ -O2 -mavx2) "
                          f"on the first {n} frames of the same stream, {passes} passes, {dt:.1f} s",
                "alignment_only": port}
    return port


def pmc_file(n_local, dens, kernel="kpe_bs"):
    """The committed rocprofv3 --pmc summary of the extraction kernel on this same workload
    (profiles/*_pmc_kpe_bs.json, produced by tools/pmc_kpe.sh + tools/pmc_to_json.py: FETCH_SIZE doubled,
    WRITE_SIZE as is — MI355X_MICROARCH.md §HBM).  (None, None) if the workload differs from the profiled
    one (10 000 frames, dens 200) or no profile is committed."""
    if n_local != FRAMES_PER_GPU or dens != DENS:
        return None, None
    import glob

    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_[a-z]_pmc_{kernel}.json")))  # (the headline stream's: <round>_<run>_pmc_<kernel>.json)
    if not files:
        return None, None
    try:
        return json.load(open(files[-1])), os.path.relpath(files[-1], ROOT)
    except Exception:
        return None, None


def time_stream(torch, frames, W, H, kmax, steps, lanes, truth=None):
    """Aligned frames/s of one more workload: `steps` blocks through AlignerPipe (as the headline), the
    extraction kernel's launch time from HIP events on the lanes' streams, its roofline fraction."""
    from remap_amd.pipeline import AlignerPipe

    n = frames.shape[0]
    pipe = AlignerPipe(W, H, kmax=kmax, lanes=lanes)
    for _ in range(2 * lanes):
        out = pipe.align(frames)
    pipe.finish()
    for lane in pipe.lanes:
        lane.ctx.set_timing(2)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = pipe.align(frames)
    nfrag = pipe.finish()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nl = [lane.ctx.kernel_launches("kpe_main") for lane in pipe.lanes]
    ms = sum(lane.ctx.kernel_ms("kpe_main") * k for lane, k in zip(pipe.lanes, nl) if k) / max(1, sum(nl))
    ktot = int(out["kcount"].sum().item())
    alg = 2.0 * W * H * n + 18.0 * ktot
    rec = {"frames": n, "size": f"{W}x{H}", "value": n * steps / dt, "unit": "frames/s", "ms_per_step": dt / steps * 1e3,
           "keypoints_per_frame": ktot / n, "fragments": nfrag, "matched_frames": int(out["matched"].sum().item()),
           "kpe_ms": ms, "kpe_frac_of_hbm_peak": alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS if ms > 0 else None}
    if truth is not None:
        dx, mt = out["dxdy"].cpu().numpy(), out["matched"].cpu().numpy()
        rec["offsets_bit_exact_vs_truth"] = bool(mt[1:].all() and (dx[1:] == truth[1:]).all())
    pipe.close()
    return rec


def profile_files(name):
    """Committed rocprofv3 summaries of workload `name` at HEAD (tools/profile_round_r04.sh): the latest of each kind."""
    import glob

    out = {}
    for kind, pat in (("kernel_stats", f"*_{name}_kernel_stats.csv"), ("pmc", f"*_{name}_pmc_*.json")):
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", pat)))
        if files:
            out[kind] = os.path.relpath(files[-1], ROOT)
    return out or None


def headline_profiles(kernel_name):
    """Committed rocprofv3 summaries behind the headline's figures (latest round in profiles/)."""
    import glob

    def last(pat):
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", pat)))
        return os.path.relpath(files[-1], ROOT) if files else None

    return {"kernel_stats_of_the_timed_region": last("r[0-9][0-9]_[a-z]_kernel_stats.csv"),
            "pmc_extraction_kernel": last(f"r[0-9][0-9]_[a-z]_pmc_{kernel_name}.json"),
            "pmc_fgm_blit_scroller": last("r[0-9][0-9]_[a-z]_blit_pmc_fgm_blit.json"),
            "drop_in_collector": last("r[0-9][0-9]_time_dropin.txt"),
            "config4_strip_occupancy_sweep": last("r[0-9][0-9]_config4_strip_sweep_640x480.jsonl")}


def other_workloads(torch, args, only=None):
    """The BASELINE.json workloads beside the headline's, one GPU, timed in this run (`only`: just that one — what
    tools/profile_round_r04.sh runs under rocprofv3 for the per-workload kernel statistics)."""
    from remap_amd.pipeline import Aligner, AlignerPipe
    from remap_amd.splice import splice
    from remap_amd.synth import rooms_v1, scroller_v1, upscale2

    res = {}
    n = args.frames
    steps = max(4, args.steps // 2)
    want = lambda name: only in (None, name)
    if want("dense"):
        fr, truth = scroller_v1(n, W, H, dens=40)
        res["dense"] = dict(time_stream(torch, torch.from_numpy(fr).cuda(), W, H, 2048, steps, args.lanes, truth),
                            workload=f"{n} frames 320x240 scroller-v1 dens 40 (configs[1], dense variant)")
        del fr
    # configs[3]: 640x480, a stream generated at that size and the pixel-doubled 320x240 stream
    if want("config4_native"):
        fr, truth = scroller_v1(n, 640, 480, dens=DENS)
        d = torch.from_numpy(fr).cuda()
        del fr
        res["config4_native"] = dict(time_stream(torch, d, 640, 480, 4096, steps, args.lanes, truth),
                                     workload=f"{n} frames 640x480 scroller-v1 generated at that size (configs[3])")
        del d
    if want("config4_doubled"):
        small, _ = scroller_v1(n, W, H, dens=DENS)
        d = torch.empty((n, 480, 640), dtype=torch.uint8, device="cuda")
        for i in range(0, n, 1000):  # (upscaled slice by slice: the host copy of the doubled stream would be 3 GB)
            d[i:i + 1000] = torch.from_numpy(upscale2(small[i:i + 1000])).cuda()
        res["config4_doubled"] = dict(time_stream(torch, d, 640, 480, 4096, steps, args.lanes),
                                      workload=f"{n} frames 640x480 = scroller-v1 pixel-doubled (configs[3] 'upscaled capture'; a doubled "
                                               "patch rarely yields keypoints — ~4 per frame — so about one frame in ten opens a new fragment)")
        # with the canvas blit: blocks in flight as in the headline's tier B (the host-side stage of block t — positions
        # to the host, ~1 000 canvas boxes — and its blit beside the alignment of block t + 1)
        pipe4 = AlignerPipe(640, 480, kmax=4096, lanes=args.lanes)
        for _ in range(2):
            pipe4.submit(d, blit=True)
        pipe4.drain()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(6):
            pipe4.submit(d, blit=True)
        pipe4.drain()
        torch.cuda.synchronize()
        res["config4_doubled"]["frames_per_s_with_blit"] = n * 6 / (time.perf_counter() - t0)
        pipe4.close()
        al = Aligner(640, 480, kmax=4096)
        al.align(d, blit=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            al.align(d, blit=True)
        torch.cuda.synchronize()
        res["config4_doubled"]["frames_per_s_with_blit_one_block_in_flight"] = n * 3 / (time.perf_counter() - t0)
        al.close()
        del d, small
    torch.cuda.empty_cache()
    for name, rec in res.items():
        rec["profiles"] = profile_files(name)
    if not want("config5"):
        return res
    # configs[4]: rooms-v1, 50 000 frames on one GPU: collect with the canvas blit, then fgs::splice
    n5 = 5 * n
    fr, _, _ = rooms_v1(n5, W, H, n_rooms=64, dens=DENS, seed=7)
    d = torch.from_numpy(fr).cuda()
    del fr
    al = Aligner(W, H, kmax=2048)
    out = al.align(d, blit=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        out = al.align(d, blit=True)
    torch.cuda.synchronize()
    t_collect = (time.perf_counter() - t0) / 3
    t0 = time.perf_counter()
    for _ in range(3):
        al.align(d)
    torch.cuda.synchronize()
    t_align = (time.perf_counter() - t0) / 3
    frags = al.fragments(d, out)
    nf = len(frags)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    spliced = splice(frags)
    torch.cuda.synchronize()
    t_splice = time.perf_counter() - t0
    res["config5"] = {"workload": f"{n5} frames 320x240 rooms-v1 (64 rooms, each visited twice): collect (kpe+kpm+frc+fgm blit), then "
                                  "fgs::splice (blend, kpe 1x1, all-pairs cellular kpm, greedy merges) — configs[4] on one GPU",
                      "frames": n5, "value": n5 / t_collect, "unit": "frames/s (collect with canvas blit)",
                      "frames_per_s_alignment_only": n5 / t_align, "fragments_collected": nf, "fragments_spliced": len(spliced),
                      "initial_pairs": nf * (nf - 1) // 2, "splice_s": t_splice,
                      "pair_matches_per_s": nf * (nf - 1) // 2 / t_splice, "profiles": profile_files("config5")}
    al.close()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--frames", type=int, default=FRAMES_PER_GPU, help="frames per GPU")
    ap.add_argument("--dens", type=int, default=DENS)
    ap.add_argument("--lanes", type=int, default=3, help="blocks in flight (AlignerPipe); 1: one stream.  Three: a lane's next block is enqueued "
                    "while its previous chain still runs beside the other lanes' extraction kernels (two: 0.453 ms per step, three: 0.430, four: 0.465)")
    ap.add_argument("--depth", type=int, default=1, help="buffer sets (contexts) per lane: 2 lets a lane's next block be enqueued before "
                    "its previous one has been settled")
    ap.add_argument("--spin-up", type=float, default=0.3, help="seconds of untimed steps after the W warm-up steps, before the "
                    "timed region: the device reaches its sustained clocks (0: none)")
    ap.add_argument("--sustain", type=float, default=1.0, help="seconds of further K-step groups after the timed one, reported as "
                    "`sustained` (median / min / max per group); 0: none")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-workloads", action="store_true", help="skip the other BASELINE.json workloads (dense, 640x480, rooms)")
    ap.add_argument("--headline-only", action="store_true", help="only the headline's steps (and the one-lane repeat): no tiers B/C, host-fed "
                                                                   "rate, copy rates or earlier kernels — for kernel traces of the timed region")
    ap.add_argument("--only-workload", default=None, help="dense | config4_native | config4_doubled | config5: time just that workload "
                                                          "and print it (for rocprofv3 runs of one workload)")
    ap.add_argument("--cpu-frames", type=int, default=10000)
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL) on a multi-GPU node; gloo only to rehearse the "
                                                      "multi-rank flow on one GPU")
    args = ap.parse_args()

    # (the pool's hosts share device memory between processes through dmabuf handles only: RCCL needs this before HIP starts)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch

    from remap_amd import lib as L
    from remap_amd.pipeline import Aligner, AlignerPipe, shard_range
    from remap_amd.synth import scroller_v1

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "gloo":
            local_rank = local_rank % max(1, torch.cuda.device_count())
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the hot path has no CPU fallback")
    L.load()
    if args.only_workload:
        print(json.dumps({"workloads": other_workloads(torch, args, only=args.only_workload)}), flush=True)
        return

    n_local = args.frames
    n_total = n_local * world
    lo, hi = shard_range(n_total, rank, world)
    # one stream for the whole job; every rank materialises only its own block
    frames_all, truth = scroller_v1(n_total, W, H, dens=args.dens) if world == 1 else (None, None)
    if world > 1:
        from remap_amd.synth import scroller_world

        wld, cam = scroller_world(n_total, W, H, args.dens)
        win = np.lib.stride_tricks.sliding_window_view(wld, W, axis=1)
        frames_host = np.ascontiguousarray(win[:, cam[lo:hi], :].transpose(1, 0, 2))
        truth = np.zeros((n_total, 2), np.int32)
        truth[1:, 0] = np.diff(cam)
    else:
        frames_host = frames_all
    dev = torch.device("cuda", local_rank)
    frames = torch.from_numpy(frames_host).to(dev)

    al = Aligner(W, H, kmax=2048, device=local_rank, dist=dist)
    # the headline: blocks taken in turn by `--lanes` lanes (streams; three by default), the small kernels of block t beside
    # the extraction of block t + 1 (AlignerPipe); `al` (one lane, the default stream) serves the
    # checks and the figures reported beside the headline
    pipe = AlignerPipe(W, H, kmax=2048, device=local_rank, dist=dist, lanes=args.lanes, depth=args.depth)
    slots = args.lanes * args.depth

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(1 if args.headline_only else args.warmup):  # (one-lane launches: one is enough for the check below)
        out = al.align(frames, n_total=n_total)
    # correctness of what is being timed: this rank's offsets, positions and fragment ids must
    # equal the generator's ground truth (one fragment, position = camera column)
    dx = out["dxdy"].cpu().numpy()
    mt = out["matched"].cpu().numpy()
    ps = out["pos"].cpu().numpy()
    fg = out["frag"].cpu().numpy()
    first = 1 if rank == 0 else 0
    cam_x = np.cumsum(truth[:, 0])
    exact = bool(mt[first:].all() and (dx[first:] == truth[lo + first:hi]).all()
                 and (ps[:, 0] == cam_x[lo:hi]).all() and (ps[:, 1] == 0).all() and (fg == 0).all()
                 and out["nfrag"] == 1)
    ktot = int(out["kcount"].sum().item())

    # the timed region records HIP events around kpe_main only (an event pair costs ~10 us of
    # stream time per launch); the other kernels are timed in a separate, untimed pass below
    # The K timed steps are enqueued back to back (defer=True: a stream of blocks, as collect
    # sees it); the host waits ONCE, inside the timed region: finish() returns the last step's
    # fragment count and raises if any of the K steps overflowed a table or a matcher limit.
    for _ in range(max(args.warmup, 2 * slots)):
        pipe.align(frames, n_total=n_total)
    pipe.finish()
    # The device needs a few hundred milliseconds of load to reach its sustained state (clocks): the first K steps
    # after idle run ~8 % slower than every later K steps of the same loop (tools/free_running_frac.py: 0.483 then
    # 0.441, 0.437 ... ms; after 0.3 s of unrelated matrix products 0.448, 0.437, 0.435).  A stream of blocks runs for
    # seconds, so the warm-up goes on, untimed, until --spin-up seconds of the same steps have passed.
    spin_steps = 0
    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < args.spin_up:
        for _ in range(2 * slots):
            pipe.align(frames, n_total=n_total)
        pipe.finish()
        spin_steps += 2 * slots
    for lane in pipe.lanes:
        lane.ctx.set_timing(2)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = pipe.align(frames, n_total=n_total)
    nfrag_timed = pipe.finish()
    barrier()
    dt = time.perf_counter() - t0
    # ... and what they computed is the ground truth too (last step's outputs)
    dx2, mt2, ps2, fg2 = (out[k].cpu().numpy() for k in ("dxdy", "matched", "pos", "frag"))
    exact = bool(exact and mt2[first:].all() and (dx2[first:] == truth[lo + first:hi]).all()
                 and (ps2[:, 0] == cam_x[lo:hi]).all() and (ps2[:, 1] == 0).all() and (fg2 == 0).all() and nfrag_timed == 1)
    # average per launch over the timed region (the timing track of kpe_mf / kpe_main), all lanes
    nl = [lane.ctx.kernel_launches("kpe_main") for lane in pipe.lanes]
    ms_kpe = sum(lane.ctx.kernel_ms("kpe_main") * k for lane, k in zip(pipe.lanes, nl) if k) / max(1, sum(nl))
    launches_per_step = sum(nl) / max(1, args.steps)
    # The same K-step groups again, back to back for --sustain seconds: the headline above is one group (the driver's
    # K and W), a few milliseconds long; this says what a stream of blocks sees over a second — median / min / max per
    # group, the extraction kernel's HIP-event time over all of them.
    sustained = None
    if args.sustain > 0:
        ngroups = int(min(400, max(3, args.sustain / max(dt, 1e-6) + 1)))
        if dist is not None:
            g = torch.tensor([ngroups], dtype=torch.int64, device=dev if args.backend != "gloo" else "cpu")
            dist.all_reduce(g, op=dist.ReduceOp.MAX)
            ngroups = int(g.item())
        for lane in pipe.lanes:
            lane.ctx.set_timing(2)
        gt = []
        for _ in range(ngroups):
            barrier()
            tg = time.perf_counter()
            for _ in range(args.steps):
                pipe.align(frames, n_total=n_total)
            pipe.finish()
            barrier()
            gt.append(time.perf_counter() - tg)
        nls = [lane.ctx.kernel_launches("kpe_main") for lane in pipe.lanes]
        ms_kpe_s = sum(lane.ctx.kernel_ms("kpe_main") * k for lane, k in zip(pipe.lanes, nls) if k) / max(1, sum(nls))
        gtt = torch.tensor(gt, dtype=torch.float64, device=dev if args.backend != "gloo" else "cpu")
        if dist is not None:
            dist.all_reduce(gtt, op=dist.ReduceOp.MAX)
        gt = sorted(float(v) for v in gtt)
        med = gt[len(gt) // 2]
        sustained = {"groups": ngroups, "steps_per_group": args.steps, "seconds": sum(gt),
                     "ms_per_step_median": med / args.steps * 1e3, "ms_per_step_min": gt[0] / args.steps * 1e3,
                     "ms_per_step_max": gt[-1] / args.steps * 1e3, "value_median": n_total * args.steps / med,
                     "kernel_ms": ms_kpe_s, "launches": int(sum(nls)),
                     "note": "the timed region repeated (barrier + synchronize around every group of K steps, max over ranks); "
                             "`value` above is ONE such group, as the driver's contract asks"}
    for lane in pipe.lanes:
        lane.ctx.set_timing(False)
    # ... and with the lanes' extraction kernels free-running (no turns): the faster arrangement, in which a launch's
    # duration no longer measures the kernel (two of them share the chip) — reported beside the headline
    dt_free = None
    if not args.headline_only and args.lanes > 1:
        pipe.set_turns(False)  # (the same lanes and streams: a fifth stream would share a hardware queue)
        for _ in range(2 * args.lanes):
            pipe.align(frames, n_total=n_total)
        pipe.finish()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            pipe.align(frames, n_total=n_total)
        pipe.finish()
        barrier()
        dt_free = time.perf_counter() - t0
        pipe.set_turns(True)
    # the same steps on ONE lane (round 2a-2e's figure), for reference.  (Not with --headline-only: the kernel
    # trace of that run then holds the timed region's launches and their warm-up, all in the same regime, and
    # its per-kernel averages are the ones this line reports.)
    dt_one, ms_kpe_one, ms_parts = dt, ms_kpe, {}
    if not args.headline_only:
        al.align(frames, n_total=n_total)
        al.ctx.set_timing(2)
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            al.align(frames, n_total=n_total, defer=True)
        al.finish()
        barrier()
        dt_one = time.perf_counter() - t0
        ms_kpe_one = al.ctx.kernel_ms("kpe_main")  # the extraction kernel with the chip to itself
        al.ctx.set_timing(True)
        for _ in range(3):
            al.align(frames, n_total=n_total)
        # per-step totals of the other kernels (kpm_build / kpm_match run several instances each)
        ms_parts = {k: al.ctx.kernel_ms(k) * al.ctx.kernel_launches(k) / 3
                    for k in ("kpe_table", "kpm_build", "kpm_match")}
    if not args.headline_only:
        # the extraction kernels of earlier rounds on this box, untimed part: the two-kernel form (rounds 1-2a:
        # kpe_main + kpe_finish) and kpe_mf (round 2; a context made with RMX_KPE_BS=0)
        os.environ["RMX_KPE_PAIR"] = "1"
        al.align(frames, n_total=n_total)
        al.ctx.set_timing(True)
        for _ in range(3):
            al.align(frames, n_total=n_total)
        pair = {k: al.ctx.kernel_ms(k) * al.ctx.kernel_launches(k) / 3 for k in ("kpe_main", "kpe_finish")}
        del os.environ["RMX_KPE_PAIR"]
        al.ctx.set_timing(False)
        os.environ["RMX_KPE_BS"] = "0"
        mf = L.Context(W, H, kmax=2048, device=local_rank)
        del os.environ["RMX_KPE_BS"]
        mf_out = mf.kpe(frames)
        mf.set_timing(True)
        for _ in range(5):
            mf.kpe(frames, *mf_out)
        ms_kpe_mf = mf.kernel_ms("kpe_main")
        mf.close()
        del mf_out

        # tier B: + fgm blit into the fragment canvases (reported beside the headline); blocks in
        # flight as above (AlignerPipe.submit / drain: the host-side stage of block t — positions to
        # the host, canvas extents — and its blit run beside the alignment of block t + 1)
        pipe.submit(frames, n_total=n_total, blit=True)
        pipe.submit(frames, n_total=n_total, blit=True)
        pipe.drain()
        barrier()
        t1 = time.perf_counter()
        nb = args.steps
        for _ in range(nb):
            pipe.submit(frames, n_total=n_total, blit=True)
        pipe.drain()
        barrier()
        dtb = time.perf_counter() - t1

        # tier C: + the frame codec main.cpp plugs into collect (nic::compress of image and median)
        pipe.submit(frames, n_total=n_total, blit=True, compress=True)
        pipe.submit(frames, n_total=n_total, blit=True, compress=True)
        pipe.drain()
        barrier()
        t3 = time.perf_counter()
        for _ in range(nb):
            pipe.submit(frames, n_total=n_total, blit=True, compress=True)
        pipe.drain()
        barrier()
        dtc = time.perf_counter() - t3
        # ... and one block at a time (rounds 1-2e's figures)
        al.align(frames, n_total=n_total, blit=True)
        barrier()
        t1 = time.perf_counter()
        nb1 = max(1, args.steps // 4)
        for _ in range(nb1):
            al.align(frames, n_total=n_total, blit=True)
        barrier()
        dtb_one = time.perf_counter() - t1

        # host-fed rate: the same step with the frames coming from pinned host
        # memory over PCIe every time — reported beside the headline, never as `value`
        pinned = torch.from_numpy(frames_host).pin_memory()
        stage = torch.empty_like(frames)
        stage.copy_(pinned, non_blocking=True)
        al.align(stage, n_total=n_total)
        barrier()
        t2 = time.perf_counter()
        nh = max(1, args.steps // 4)
        for _ in range(nh):
            stage.copy_(pinned, non_blocking=True)
            al.align(stage, n_total=n_total)
        barrier()
        dth = time.perf_counter() - t2
        # what a plain device copy of the same footprint reaches on this box (the practical HBM
        # ceiling next to the 8 TB/s spec figure): read n*W*H + write n*W*H bytes, with the library's
        # own streaming copy kernel (rmx_copy) and, for reference, torch's device copy
        def copy_rate(fn):
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            fn()
            ev0.record()
            for _ in range(10):
                fn()
            ev1.record()
            torch.cuda.synchronize()
            return 2.0 * frames.numel() * 10 / (ev0.elapsed_time(ev1) * 1e-3) / 1e9

        copy_gbs = copy_rate(lambda: al.ctx.copy(stage, frames))
        torch_copy_gbs = copy_rate(lambda: stage.copy_(frames))
        del pinned, stage
    else:  # (tools/profile_round.sh: the kernel trace of this run holds the headline's launches only)
        pair = {"kpe_main": 0.0, "kpe_finish": 0.0}
        ms_kpe_mf = 0.0
        dtb = dtc = dth = dtb_one = float("inf")
        nb = nb1 = nh = 1
        copy_gbs = torch_copy_gbs = None

    tmax = torch.tensor([dt, dtb, dth, dtc, 0.0 if exact else 1.0, dt_one, dtb_one, dt_free or 0.0], dtype=torch.float64, device=dev if args.backend != "gloo" else "cpu")
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt, dtb, dth, dtc, bad, dt_one, dtb_one, dt_free = (float(v) for v in tmax)
    exact = bad == 0.0  # on every rank

    if rank == 0:
        # algorithmic bytes of one kpe_main launch: SURVEY.md §8d's per-frame figure
        # (2*W*H + 18*K) x the frames one launch processes
        alg_bytes = (2.0 * W * H * n_local + 18.0 * ktot) / max(1.0, launches_per_step)
        achieved = alg_bytes / (ms_kpe * 1e-3) / 1e9 if ms_kpe > 0 else 0.0
        kernel_name = pipe.lanes[0].ctx.kpe_kernel()
        pmc, pmc_src = pmc_file(n_local, args.dens, kernel_name) if launches_per_step == 1 else (None, None)
        traffic = float(pmc["hbm_bytes_per_launch"]) if pmc else None
        valu = float(pmc["counters_mean_per_launch"]["SQ_INSTS_VALU"]) if pmc else None
        line = {
            "metric": "aligned frames/s, 320x240 stream (kpe+kpm+frc: alignment only, no canvas blit), offsets bit-exact",
            "value": n_total * args.steps / dt,
            "unit": "frames/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": f"{n_local} frames 320x240 scroller-v1 (dens {args.dens}) per GPU, "
                                   f"kpe+kpm+frc HIP kernels, inputs resident in HBM",
                       "frames_total": n_total, "keypoints_per_frame": ktot / n_local,
                       "parallelism": f"frame-sharded x{world}" + (" + RCCL all-gather of boundary tables and 16-B scan summaries"
                                                                   if world > 1 else "")
                                      + (f", {args.lanes} blocks in flight per GPU (the small kernels of block t beside the extraction of block t+1)"
                                         if args.lanes > 1 else "")},
            "offsets_bit_exact_vs_truth": exact,
            "profiles": headline_profiles(kernel_name),
            "spin_up": {"seconds": args.spin_up, "untimed_steps": spin_steps,
                        "note": "untimed steps of the same loop between the W warm-up steps and the K timed ones: the device's "
                                "sustained state (the first K steps after idle run ~8 % slower than all later ones)"},
            "sustained": (dict(sustained, roofline_frac=alg_bytes / (sustained["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
                               if sustained["kernel_ms"] > 0 else None) if sustained else None),
            "frames_per_s_one_block_in_flight": n_total * args.steps / dt_one,
            "frames_per_s_extraction_kernels_free_running": (n_total * args.steps / dt_free) if dt_free else None,
            "frames_per_s_with_blit": n_total * nb / dtb,
            "frames_per_s_with_blit_one_block_in_flight": n_total * nb1 / dtb_one,
            "frames_per_s_with_blit_and_codec": n_total * nb / dtc,
            "frames_per_s_host_fed": n_total * nh / dth,
            "kernel_ms": dict({kernel_name: ms_kpe}, **ms_parts),
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "traffic_source": (pmc_src + " (rocprofv3 --pmc FETCH_SIZE x 2 + WRITE_SIZE of this kernel on this workload, "
                                            "a committed run: counters cannot be collected from inside the process)") if pmc_src else None,
                         "kernel": kernel_name + " (whole keypoint extraction: " + KERNEL_NOTES.get(kernel_name, "") + "; descriptor rows are ordered by kpe_table, 0.03 ms)",
                         "measured_in_this_run": ["achieved", "frac", "one_block_in_flight", "previous_kernels", "device_copy_gbs",
                                                  "torch_device_copy_gbs", "algorithmic_bytes_per_launch"],
                         "one_block_in_flight": {"note": "the same kernel with the chip to itself (single stream, the K steps timed again); "
                                                         "with several blocks in flight the small kpm kernels of the previous blocks run beside it",
                                                 "kernel": al.ctx.kpe_kernel(), "kernel_ms": ms_kpe_one,
                                                 "frac": alg_bytes / (ms_kpe_one * 1e-3) / 1e9 / HBM_PEAK_GBS if ms_kpe_one > 0 else None},
                         "previous_kernels": {"note": "same box, same frames, same algorithmic bytes",
                                              "kpe_mf_ms": ms_kpe_mf,
                                              "kpe_mf_frac": alg_bytes / (ms_kpe_mf * 1e-3) / 1e9 / HBM_PEAK_GBS if ms_kpe_mf > 0 else None,
                                              "kpe_main_ms": pair["kpe_main"], "kpe_finish_ms": pair["kpe_finish"],
                                              "kpe_main_plus_finish_frac": alg_bytes / ((pair["kpe_main"] + pair["kpe_finish"]) * 1e-3) / 1e9 / HBM_PEAK_GBS
                                              if pair["kpe_main"] > 0 else None},
                         "device_copy_gbs": copy_gbs, "frac_of_device_copy": achieved / copy_gbs if copy_gbs else None,
                         "torch_device_copy_gbs": torch_copy_gbs,
                         "valu_wave_instructions": valu, "valu_wave_instructions_source": pmc_src if valu else None,
                         "algorithmic_bytes_per_launch": alg_bytes},
        }
        if world == 1 and not args.no_cpu:
            cb = cpu_baseline(frames_host[: min(args.cpu_frames, n_local)])
            line["cpu_baseline"] = cb
            # the same stages on both sides (the headline has no blit, the reference's collect does)
            line["like_for_like"] = {
                "with_blit": {"gpu_frames_per_s": n_total * nb / dtb, "cpu_frames_per_s": cb["value"] if cb["kind"] == "reference" else None,
                              "cpu_kind": "reference"},
                "alignment_only": {"gpu_frames_per_s": n_total * args.steps / dt,
                                   "cpu_frames_per_s": (cb.get("alignment_only") or cb)["value"], "cpu_kind": "port"}}
        else:
            line["cpu_baseline"] = None
        if world == 1 and not args.no_workloads:
            al.close()
            pipe.close()
            del frames
            torch.cuda.empty_cache()
            line["workloads"] = other_workloads(torch, args)
        print(json.dumps(line), flush=True)
    if not (rank == 0 and world == 1 and not args.no_workloads):
        al.close()
        pipe.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
