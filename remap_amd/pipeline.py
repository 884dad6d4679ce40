"""Batched alignment pipeline — the host side of frc::collector::collect (reference
frc.hpp:55-127) on top of the C ABI, for one GPU or for a frame-sharded group of GPUs.

Reference loop (sequential, frc.hpp:97-122)      this driver (batched, per rank)
  extract(frame t)                                rmx_kpe_batch(all local frames)
  match(grid t-1, grid t)                         rmx_kpm_seq (+ halo table from rank-1)
  position += offset | new fragment               rmx_frc_scan over the gathered offsets
  fragment.blit(position, frame)                  rmx_fgm_blit_batch per fragment canvas

Multi-GPU: the stream is cut into contiguous frame blocks, one per rank.
kpe needs no communication.  kpm needs, per rank, the descriptor table of the frame just
before its block: the boundary tables (last frame of every rank, kmax*18 B + count) are
exchanged with ONE all-gather (RCCL over xGMI on GPUs, gloo in the CPU tests).  Positions
and fragment ids are a segmented scan of the per-frame offsets: every rank scans its own
block and the blocks are stitched with a 16-byte summary per rank (second tiny all-gather),
so a rank never touches another rank's per-frame data to align its frames.  Whole-stream
positions (12 B/frame) are gathered only to size fragment canvases.
"""
from __future__ import annotations

import numpy as np

from . import lib as L

__all__ = ["shard_range", "Aligner", "AlignerPipe", "StepResult", "gather_boundary_tables", "gather_offsets", "sum_subcanvases", "sum_bins_mod_2_16",
           "fold_block_summaries", "gather_block_rows", "canvas_boxes"]


class StepResult(dict):
    """What one step of the hot path returns: device tensors of the local block by name.

    A deferred step (`Aligner.align(defer=True)`, `AlignerPipe.align`) is PROVISIONAL until it has been settled:
    the matcher may have left tickets to `rmx_kpm_resolve` (a region beyond its LDS instances, an instance outside
    the lane's launch set), and settling rewrites offsets, positions and fragment ids.  Those entries are
    therefore not readable before `settle()` / `finish()` — reading them after a bare `synchronize()` raises
    instead of handing out numbers that may still change.  The extraction's outputs (median, desc, kcount) are
    never rewritten and stay readable."""

    PROVISIONAL = frozenset(("dxdy", "matched", "pos", "frag", "pos_all", "frag_all", "canvases"))  # (`nfrag` is None until then)

    def __init__(self, *a, settled=True, **kw):
        super().__init__(*a, **kw)
        self.settled = settled

    def _guard(self, key):
        if not self.settled and key in self.PROVISIONAL:
            raise RuntimeError(f"'{key}' of a deferred step is provisional until the step has been settled "
                               "(AlignerPipe.settle(out) / finish(), Aligner.finish())")

    def __getitem__(self, key):
        self._guard(key)
        return dict.__getitem__(self, key)

    def get(self, key, default=None):
        self._guard(key)
        return dict.get(self, key, default)

    def raw(self, key):
        """The entry whatever the state (the pipeline's own use while it settles the step)."""
        return dict.__getitem__(self, key)


def shard_range(n_total: int, rank: int, world: int):
    """Contiguous block [lo, hi) of frames owned by `rank` (blocks differ by at most one)."""
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _all_gather_flat(flat, world, dist, group):
    """all_gather_into_tensor of a 1-D/2-D tensor along dim 0.  RCCL gathers device tensors
    in place; the gloo backend (CPU tests, single-GPU rehearsals) gets a host round trip."""
    import torch

    out_shape = (world * flat.shape[0],) + tuple(flat.shape[1:])
    if dist.get_backend(group) == "gloo":
        host = flat.cpu() if flat.is_cuda else flat.contiguous()
        out = torch.empty(out_shape, dtype=flat.dtype)
        dist.all_gather_into_tensor(out, host, group=group)
        return out.to(flat.device)
    # RCCL gathers in place on the device: one contiguous device buffer per rank, same dtype and size everywhere
    assert flat.is_cuda, "the nccl (RCCL) backend gathers device tensors"
    assert flat.dtype in (torch.uint8, torch.int32, torch.int16, torch.int64), flat.dtype
    flat = flat.contiguous()
    out = torch.empty(out_shape, dtype=flat.dtype, device=flat.device)
    assert out.is_contiguous() and out.numel() == world * flat.numel()
    dist.all_gather_into_tensor(out, flat, group=group)
    return out


def gather_boundary_tables(last_desc, last_k, dist=None, group=None):
    """All-gather the last frame's descriptor table of every rank.

    last_desc: (kmax, 18) uint8 tensor, last_k: (1,) int32 tensor (device of the backend).
    Returns (tables [world, kmax, 18], counts [world]).  Without torch.distributed
    (single process) returns the inputs with a leading axis."""
    if dist is None or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return last_desc[None], last_k.reshape(1)
    import torch

    world = dist.get_world_size(group)
    # one flat buffer per rank: [table bytes | count (4 bytes, little endian)] — the table first,
    # so that every rank's table sits 4-byte aligned in the gathered buffer and can be used in
    # place (a view, no copy) as the halo of rmx_kpm_seq
    nb = last_desc.numel()
    pad = (-nb) % 4
    flat = torch.empty(nb + pad + 4, dtype=torch.uint8, device=last_desc.device)
    flat[:nb].copy_(last_desc.reshape(-1))
    flat[nb + pad:].copy_(last_k.reshape(1).to(torch.int32).view(torch.uint8))
    allf = _all_gather_flat(flat, world, dist, group).view(world, -1)
    cnts = allf[:, nb + pad:].view(torch.int32).reshape(world)  # (row stride is a multiple of 4)
    tabs = allf[:, :nb].view((world,) + tuple(last_desc.shape))
    return tabs, cnts


def gather_offsets(dxdy, matched, counts, dist=None, group=None):
    """All-gather per-frame (dx, dy, matched) of every rank's block into stream order.
    counts: list of block sizes per rank (blocks may differ by one -> padded gather)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return dxdy, matched
    import torch

    world = dist.get_world_size(group)
    m = max(counts)
    pack = torch.zeros((m, 3), dtype=torch.int32, device=dxdy.device)
    pack[: dxdy.shape[0], :2] = dxdy
    pack[: dxdy.shape[0], 2] = matched.to(torch.int32)
    allp = _all_gather_flat(pack, world, dist, group).view(world, m, 3)
    parts = [allp[r, : counts[r]] for r in range(world)]
    cat = torch.cat(parts)
    return cat[:, :2].contiguous(), cat[:, 2].to(torch.uint8).contiguous()


def fold_block_summaries(summaries, rank):
    """Carry into rank `rank`'s block from the scan summaries of the blocks before it.

    summaries: (world, 4) integers per block [has a fragment start, x, y, starts]: position of
    the block's last frame relative to its last fragment start (or to the block's begin if it
    has none), and the number of fragment starts in it (frc.hpp:109-115).  Returns
    ((x, y), fragments) = position and fragment id of the frame just before the block."""
    ax = ay = ac = 0
    for q in range(rank):
        r, x, y, c = (int(v) for v in summaries[q])
        r &= 1  # (bit 1: the block has tickets left to rmx_kpm_resolve)
        ax, ay = (x, y) if r else (ax + x, ay + y)
        ac += c
    return (ax, ay), ac


def gather_block_rows(rows, counts, dist, group=None):
    """All-gather per-frame rows (n_local, k) int32 of every rank's block into stream order
    (blocks may differ by one frame -> padded gather)."""
    import torch

    world = dist.get_world_size(group)
    m = max(counts)
    pack = torch.zeros((m, rows.shape[1]), dtype=torch.int32, device=rows.device)
    pack[: rows.shape[0]] = rows
    allp = _all_gather_flat(pack, world, dist, group).view(world, m, rows.shape[1])
    return torch.cat([allp[r, : counts[r]] for r in range(world)])


def sum_subcanvases(sub, zero, gbox, dist, group=None, device=None):
    """Bin-wise sum modulo 2^16 of every rank's sub-canvas of ONE fragment (fgm.hpp:99-113).

    sub: (ch, cw, 16) int16 tensor or None (no frames of the fragment on this rank); zero:
    canvas coordinates of its top-left; gbox = (gw, gh, gzx, gzy): the fragment's whole box.
    RCCL has no 16-bit integer type, so the sub-canvases are widened to int32, summed with
    ONE all-reduce and narrowed again — exact: the int32 sum of at most `world` uint16 values
    cannot overflow and its low 16 bits are the wrapped sum.  Returns (gh, gw, 16) int16."""
    import torch as t

    gw, gh, gzx, gzy = gbox
    gloo = dist.get_backend(group) == "gloo"
    acc = t.zeros((gh, gw, 16), dtype=t.int32, device="cpu" if gloo else device)
    if sub is not None:
        ch, cw = sub.shape[0], sub.shape[1]
        y0, x0 = zero[1] - gzy, zero[0] - gzx
        acc[y0:y0 + ch, x0:x0 + cw] += (sub.to(t.int32) & 0xffff).to(acc.device)
    dist.all_reduce(acc, op=dist.ReduceOp.SUM, group=group)
    whole = (acc & 0xffff).to(t.int16)  # int32 -> int16 keeps the low 16 bits
    return whole if device is None else whole.to(device)


def sum_bins_mod_2_16(flat, dist=None, group=None):
    """In place: the uint16 bins of `flat` (an int16 tensor, every rank's own counts) summed over the ranks modulo
    2^16 (fgm::dot_type wraps, fgm.hpp:14).  The bins travel as int32 — RCCL has no 16-bit integer type; the sum of
    at most `world` values below 2^16 cannot overflow — and are narrowed again.  Without a process group (or with
    one rank) the widening and narrowing still run: the identity, which the tests pin on device tensors."""
    import torch as t

    assert flat.dtype == t.int16 and flat.is_contiguous()
    wide = flat.to(t.int32) & 0xffff
    if dist is not None and dist.is_initialized() and dist.get_world_size(group) > 1:
        gloo = dist.get_backend(group) == "gloo"
        if gloo:
            wide = wide.cpu()
        else:
            assert wide.is_cuda and wide.is_contiguous(), "the nccl (RCCL) backend reduces device tensors"
        dist.all_reduce(wide, op=dist.ReduceOp.SUM, group=group)
    flat.copy_((wide & 0xffff).to(t.int16))  # int32 -> int16 keeps the low 16 bits
    return flat


def canvas_boxes(pos, starts, W, H):
    """Canvas boxes of fragments that are runs of the stream: pos (n, 2) integer positions, starts = first
    frame of every run.  Returns (w, h, zero_x, zero_y) arrays.  fgm.hpp:190-233 grows the canvas by whole
    steps (W, H) on the side a blit sticks out of, starting from one step at the origin; the result does
    not depend on the order of the blits: left edge = -(smallest multiple of W covering -min x), right edge
    = W + (smallest multiple covering max x), and the same in y — what rmx_fgm_extent replays frame by frame."""
    pos = np.asarray(pos, np.int64)
    mn = np.minimum.reduceat(pos, starts, axis=0)
    mx = np.maximum.reduceat(pos, starts, axis=0)
    step = np.array([W, H], np.int64)
    ceil_steps = lambda v: -(-v // step) * step  # smallest multiple of the step covering v >= 0
    zero = -ceil_steps(np.maximum(-mn, 0))
    far = step + ceil_steps(np.maximum(mx, 0))
    dims = far - zero
    return dims[:, 0], dims[:, 1], zero[:, 0], zero[:, 1]


class Aligner:
    """frc::collector for a batch of frames resident in HBM (one rank's block)."""

    def __init__(self, width, height, kmax=2048, device=0, dist=None, group=None):
        import torch

        self.torch = torch
        self.ctx = L.Context(width, height, grid=(4, 2), overlap=16, kmax=kmax, device=device)
        self.ctx.kpm_adaptive(True)  # (_check_step resolves flag 128, which is all the adaptive launch set asks for)
        self.W, self.H, self.kmax = width, height, kmax
        self.dist = dist if (dist is not None and dist.is_initialized()) else None
        self.group = group
        self.rank = self.dist.get_rank(group) if self.dist else 0
        self.world = self.dist.get_world_size(group) if self.dist else 1
        self._bufs = {}
        self.blit_gate = None
        self._deferred = None   # (out, halo) of a deferred step whose status has not been looked at yet
        self._last_nfrag = None

    def _buf(self, name, shape, dtype):
        t = self.torch
        b = self._bufs.get(name)
        if b is None or tuple(b.shape) != tuple(shape) or b.dtype != dtype:
            b = t.empty(shape, dtype=dtype, device=self.ctx.device)
            self._bufs[name] = b
        return b

    def align(self, frames, n_total=None, blit=False, check=True, compress=False, defer=False):
        """One pass of the hot path over this rank's block `frames` (n, H, W) uint8 on device.

        Returns a dict with device tensors for the LOCAL block (frames lo..hi of the stream):
        median, desc, kcount, dxdy, matched and pos, frag in GLOBAL canvas coordinates /
        fragment numbering, plus `lo`, `hi`, `nfrag` (fragments of the whole stream).  With
        several ranks the global values come from a 16-byte scan summary per rank (ONE tiny
        all-gather), not from gathering every frame's offset; `pos_all` / `frag_all` (whole
        stream, identical on all ranks) are gathered only when canvases are asked for.  If
        blit=True, `canvases` for the fragments that have frames on this rank; if compress=True, `nic_image` and `nic_median`: the
        nic::compress blobs of every local frame and median as (blobs (n, stride), sizes (n,))
        device tensors — what collect stores with each frame (frc.hpp:118) behind main.cpp's
        codec.

        defer=True enqueues the step and returns without waiting for it (`nfrag` is None, no status
        check): a stream of blocks is aligned back to back and the host waits once, in
        `finish()`, which returns the fragment count of the last step and raises if ANY of the
        deferred steps overflowed a table or a matcher limit (the status words are sticky)."""
        t = self.torch
        ctx = self.ctx
        self._settle()  # a deferred step of this Aligner: its buffers and work lists are about to be reused
        n = frames.shape[0]
        if n_total is None:
            n_total = n * self.world
        lo, hi = shard_range(n_total, self.rank, self.world)
        assert hi - lo == n, "block size does not match shard_range"
        median = self._buf("median", frames.shape, t.uint8)
        desc = self._buf("desc", (n, self.kmax, L.DESC_BYTES), t.uint8)
        kcount = self._buf("kcount", (n,), t.int32)
        ctx.kpe(frames, median, desc, kcount)  # (AlignerPipe: the extraction kernels of its lanes take turns, rmx_kpe_take_turns)
        halo = None
        if self.world > 1:
            tabs, cnts = gather_boundary_tables(desc[n - 1], kcount[n - 1:n], self.dist, self.group)
            if self.rank > 0:
                halo = (tabs[self.rank - 1].contiguous(), cnts[self.rank - 1:self.rank].contiguous())
        dxdy, matched = ctx.kpm_seq(desc, kcount, halo=halo)
        if self.world == 1 and not defer:
            pos, frag, carry = ctx.frc_scan(dxdy, matched, first_is_start=True)  # synchronises
            nfrag = carry[1] + 1
        elif self.world == 1:
            pos, frag, mine = ctx.frc_scan_block(dxdy, matched, first_is_start=True)  # (leaves the fragment count for finish())
            nfrag = None
        else:
            # every rank scans its block on its own; the blocks' 16-byte summaries (did a fragment
            # start in it, where its last frame sits relative to that start or to the block's
            # begin, how many starts) are exchanged and folded ON THE DEVICE: scan, all-gather and
            # fix-up are enqueued back to back, the host waits once
            pos, frag, mine = ctx.frc_scan_block(dxdy, matched, first_is_start=(self.rank == 0))
            summ = _all_gather_flat(mine, self.world, self.dist, self.group)
            ctx.frc_apply_blocks(summ.contiguous(), self.rank, self.world, pos, frag)
            nfrag = None if defer else ctx.frc_finish()
        out = StepResult(median=median, desc=desc, kcount=kcount, dxdy=dxdy, matched=matched, pos=pos, frag=frag,
                         lo=lo, hi=hi, nfrag=nfrag, n_total=n_total, settled=not defer)
        if defer:
            assert not blit, "canvases need the positions on the host: not with defer=True"
            assert check, "a deferred step is checked when it is settled"
            # settled (status words, tickets left to rmx_kpm_resolve) before this Aligner's next step
            # reuses the buffers and the matcher's work list, or in finish()
            self._deferred = (out, halo)
            if compress:
                out["nic_image"] = ctx.nic_compress(frames.reshape(n, -1), pack=False)
                out["nic_median"] = ctx.nic_compress(median.reshape(n, -1), pack=False)
            return out
        if check:
            self._check_step(out, halo)
        if compress:
            out["nic_image"] = ctx.nic_compress(frames.reshape(n, -1), pack=False)
            out["nic_median"] = ctx.nic_compress(median.reshape(n, -1), pack=False)
            if check:
                ctx._chk(ctx.lib.rmx_nic_status(ctx.h), "rmx_nic_status")
        if blit:
            self.gather_positions(out)
            out["canvases"] = self.blit(frames, out["pos_all"], out["frag_all"], lo, hi)
        return out

    def _check_step(self, out, halo):
        """Status words of the step that produced `out` (the stream has been waited for: rmx_frc_scan or
        rmx_frc_finish cached them).  Raises on an overflow — on EVERY rank of a sharded stream if one rank
        has one (the failure bit travels with the scan summaries), so that no rank is left alone in a
        collective; runs the matcher's global-memory instance for tickets the LDS instances left
        (rmx_kpm_resolve) and derives the positions again, all ranks together."""
        ctx = self.ctx
        rc, _ = ctx.kpe_status(want_max=False)
        kpe_err = None
        if rc:
            _, maxk = ctx.kpe_status()
            kpe_err = L.RemapError(rc, "rmx_kpe_status", f"a frame of this step has {maxk} keypoints > kmax={self.kmax}")
        rc, fl = ctx.kpm_status()
        if kpe_err is not None:
            raise kpe_err
        if fl & ~(128 | L.RMX_KPM_REMOTE_FAILURE):
            raise L.RemapError(L.RMX_ERR_CAPACITY, "rmx_kpm_status", f"matcher workspace limit, flags={fl}")
        if fl & L.RMX_KPM_REMOTE_FAILURE:
            raise L.RemapError(L.RMX_ERR_CAPACITY, "rmx_frc_finish", "another rank's block overflowed a table or a matcher limit")
        if fl & 128:
            # regions / pairs the LDS matcher left to its global-memory instance (a frame with
            # thousands of keypoints in one region): tickets computed there, offsets rewritten,
            # positions derived again.  On a sharded stream every rank gets here if one does
            # (the flag travels with the scan summaries), so the collectives below pair up.
            desc, kcount, dxdy, matched = out.raw("desc"), out.raw("kcount"), out.raw("dxdy"), out.raw("matched")
            # (a hard limit inside the redone pass raises here on this rank; the other ranks learn of it from the
            # failure bit of the summaries gathered below, so no rank is left alone in a collective)
            err = None
            try:
                ctx.kpm_resolve(desc, kcount, dxdy, matched, halo=halo)
            except L.RemapError as e:
                if self.world == 1:
                    raise
                err = e
                ctx.lib.rmx_debug_poke_status(ctx.h, 1, 4)  # the sticky word the scan summary reads its failure bit from
            if self.world == 1:
                pos, frag, carry = ctx.frc_scan(dxdy, matched, first_is_start=True)
                nfrag = carry[1] + 1
                _, fl2 = ctx.kpm_status()
            else:
                pos, frag, mine = ctx.frc_scan_block(dxdy, matched, first_is_start=(self.rank == 0))
                summ = _all_gather_flat(mine, self.world, self.dist, self.group)
                ctx.frc_apply_blocks(summ.contiguous(), self.rank, self.world, pos, frag)
                nfrag = ctx.frc_finish()
                _, fl2 = ctx.kpm_status()  # (128: the flag of the first pass, gathered again)
            if err is not None:
                raise err
            if fl2 & ~128:  # the redone pass is the only one that built the frames its first pass skipped
                where = "another rank's block" if fl2 & L.RMX_KPM_REMOTE_FAILURE and not fl2 & ~(128 | L.RMX_KPM_REMOTE_FAILURE) else "this block"
                raise L.RemapError(L.RMX_ERR_CAPACITY, "rmx_kpm_resolve", f"matcher workspace limit in the redone step ({where}), flags={fl2}")
            out.update(pos=pos, frag=frag, nfrag=nfrag)

    def _settle(self):
        """Complete the deferred step, if there is one: wait for it, check it, resolve it."""
        if self._deferred is None:
            return self._last_nfrag
        out, halo = self._deferred
        self._deferred = None
        out["nfrag"] = self.ctx.frc_finish()
        self._check_step(out, halo)
        out.settled = True  # offsets, positions and fragment ids are final from here on
        self._last_nfrag = out["nfrag"]
        return self._last_nfrag

    def finish(self):
        """Wait for the deferred step: its fragment count; raises if it overflowed a table or a matcher
        limit (every deferred step is settled before the next one starts on the same Aligner, so the
        error names the step it belongs to)."""
        return self._settle()

    def gather_positions(self, out):
        """pos_all / frag_all: positions and fragment ids of the WHOLE stream on every rank
        (12 B per frame, one all-gather) — needed to size fragment canvases, not to align."""
        if "pos_all" in out:
            return
        if self.world == 1:
            out["pos_all"], out["frag_all"] = out["pos"], out["frag"]
            return
        t = self.torch
        n_total = out["n_total"]
        counts = [shard_range(n_total, r, self.world)[1] - shard_range(n_total, r, self.world)[0]
                  for r in range(self.world)]
        rows = t.cat([out["pos"], out["frag"].to(t.int32)[:, None]], dim=1).contiguous()
        allr = gather_block_rows(rows, counts, self.dist, self.group)
        out["pos_all"], out["frag_all"] = allr[:, :2].contiguous(), allr[:, 2].contiguous()

    def blit(self, frames, pos, frag, lo, hi):
        """fgm::fragment::blit of this rank's frames into the canvases of ALL fragments of the stream.

        The canvas box of a fragment (zero, dims) follows from all its positions (fgm.hpp:190-233: the
        smallest box of whole steps around the frame at (0, 0) that covers every blit; `canvas_boxes`
        is that rule in closed form, vectorised over the fragments).  Fragments of equal canvas height
        are laid side by side in ONE wide canvas (64-pixel aligned), so a block is blitted with one
        zeroed allocation and one rmx_fgm_blit_batch launch per distinct height — the column bins of the
        kernel then separate the fragments by themselves — instead of an extent call, a memset and a
        launch per fragment (a stream with hundreds of fragments spent its time in that host loop).
        Returns {fragment: (canvas, zero, (w, h, zero))}: `canvas` is a (h, w, 16) int16 view into the
        stacked buffer (rows are strided; `.contiguous()` for the C ABI).  On a sharded stream every rank
        holds the same layout and has counted only its own frames: the bin-wise sum modulo 2^16 over the
        ranks of the one flat buffer (`sum_canvases`) is the reference's canvas (fgm.hpp:99-113)."""
        t = self.torch
        W, H = self.W, self.H
        # positions and fragment ids in ONE copy to the host (one wait instead of two)
        both = t.cat([pos, frag.to(t.int32)[:, None]], dim=1).cpu().numpy()
        pos_h, frag_h = both[:, :2].astype(np.int64), both[:, 2]
        # a fragment's frames are one run of the stream (frc.hpp:97-115: ids count the cuts)
        cuts = np.flatnonzero(frag_h[1:] != frag_h[:-1]) + 1
        starts = np.concatenate([[0], cuts]).astype(np.int64)
        fids = frag_h[starts]
        assert len(np.unique(fids)) == len(fids), "fragment ids are not runs of the stream"
        gw, gh, gzx, gzy = canvas_boxes(pos_h, starts, W, H)
        # classes of equal height, fragments side by side
        heights, cls_of = np.unique(gh, return_inverse=True)
        wpad = -(-gw // 64) * 64
        xoff = np.zeros(len(fids), np.int64)
        wtot = np.zeros(len(heights), np.int64)
        for c in range(len(heights)):  # (a handful of classes; the fragments inside one are vectorised)
            m = cls_of == c
            xoff[m] = np.cumsum(wpad[m]) - wpad[m]
            wtot[c] = int(wpad[m].sum())
        dots = heights.astype(np.int64) * wtot
        base = np.concatenate([[0], np.cumsum(dots)])
        flat = self._bufs.get("canvas_flat")
        if flat is None or flat.numel() != int(base[-1]) * 16:
            flat = t.empty(int(base[-1]) * 16, dtype=t.int16, device=self.ctx.device)
            self._bufs["canvas_flat"] = flat
        if self.blit_gate is not None:  # (AlignerPipe: the blit kernels wait for the other lane's extraction)
            self.blit_gate()
        # canvases without frames of this rank (and blocks too long for one launch) are zeroed; the others are
        # written in full by the blit itself (rmx_fgm_blit_fresh: no memset, the canvas is not read)
        fresh = (hi - lo) <= 65535
        stacked = [flat[int(base[c]) * 16:int(base[c + 1]) * 16].view(int(heights[c]), int(wtot[c]), 16) for c in range(len(heights))]
        # this rank's frames: position inside the stacked canvas of their class, and the class
        run_of = np.searchsorted(starts, np.arange(lo, hi), side="right") - 1
        adj = np.empty((hi - lo, 3), np.int32)
        adj[:, 0] = pos_h[lo:hi, 0] - gzx[run_of] + xoff[run_of]
        adj[:, 1] = pos_h[lo:hi, 1] - gzy[run_of]
        adj[:, 2] = cls_of[run_of]
        dev = t.from_numpy(adj).to(self.ctx.device)
        apos, acls = dev[:, :2].contiguous(), dev[:, 2].contiguous()
        mine = set(int(c) for c in np.unique(adj[:, 2]))
        if not fresh:
            flat.zero_()
        for c in range(len(heights)):
            if c in mine:
                self.ctx.fgm_blit(frames, apos, stacked[c], (0, 0), frag_id=acls, fragment=c, fresh=fresh)
            elif fresh:
                stacked[c].zero_()
        canvases = {}
        for i, fid in enumerate(fids):
            c = int(cls_of[i])
            box = (int(gw[i]), int(gh[i]), (int(gzx[i]), int(gzy[i])))
            canvases[int(fid)] = (stacked[c][:, int(xoff[i]):int(xoff[i]) + int(gw[i])], box[2], box)
        self._layout = (flat, fids, starts)
        return canvases

    def sum_canvases(self):
        """The stacked canvases of the last `blit`, summed over the ranks in place: ONE all-reduce for all
        fragments.  RCCL has no 16-bit integer type, so the bins travel as int32 (the sum of at most
        `world` uint16 values cannot overflow) and are narrowed again — exact modulo 2^16."""
        if self.world == 1:
            return
        sum_bins_mod_2_16(self._layout[0], self.dist, self.group)

    def fragments(self, frames, out):
        """collector.complete() for a (sharded) stream: the whole canvas of every fragment on every
        rank, as splice.Fragment objects (normalised: zero = (0, 0), fgm.hpp:137-143)."""
        from .splice import Fragment

        self.gather_positions(out)
        canv = out.get("canvases")
        if canv is None:
            canv = self.blit(frames, out["pos_all"], out["frag_all"], out["lo"], out["hi"])
        self.sum_canvases()
        pos_h = out["pos_all"].cpu().numpy()
        _, fids, starts = self._layout
        ends = np.concatenate([starts[1:], [len(pos_h)]])
        result = []
        for fid, a, b in zip(fids, starts, ends):
            canvas, (gzx, gzy), _ = canv[int(fid)]
            frames_list = [(int(i), (int(pos_h[i, 0]) - gzx, int(pos_h[i, 1]) - gzy)) for i in range(int(a), int(b))]
            result.append(Fragment(canvas.contiguous(), (0, 0), (self.W, self.H), frames_list))
        return result

    def close(self):
        self.ctx.close()


def lane_groups(dist, group, lanes):
    """One process group per lane over the ranks of `group` (collective: every rank calls it, same arguments)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size(group) == 1 or lanes == 1:
        return [group] * lanes
    ranks = dist.get_process_group_ranks(group) if group is not None else list(range(dist.get_world_size()))
    backend = dist.get_backend(group)
    return [dist.new_group(ranks=ranks, backend=backend) for _ in range(lanes)]


class AlignerPipe:
    """Blocks of a stream in flight on `lanes` Aligners (each with its own HIP stream and
    workspaces), taken in turn.

    A step of the hot path is one chip-filling kernel (the extraction: kpe_pc / kpe_bs_ring / kpe_bs, three
    quarters of the step's instructions) followed by a chain of small latency-bound ones (table, kpm_build*,
    kpm_match*, declare, scan) that leave most of the chip idle.  With several lanes (bench.py: three) the chain
    of block t runs beside the extraction of block t + 1.  The extraction kernels themselves take turns (rmx_kpe_take_turns: an event
    right behind each extraction kernel): two of them side by side only share the chip.

    Every block is a step of its own, as with `Aligner.align(defer=True)`: its results are final once it
    has been settled — by `settle(out)`, by its lane's next block, or by `finish()`, which waits for
    all lanes and returns the fragment count of the last block (raises if any block overflowed).
    With several ranks every lane gets a process group of its own (the same ranks as `group`, made in the same
    order on every rank): collectives of ONE group run in issue order on the group's internal stream, so with a
    shared group lane B's boundary all-gather — which waits for lane B's extraction kernel — would sit in front
    of lane A's scan-summary all-gather and the lanes would serialise."""

    def __init__(self, width, height, kmax=2048, device=0, dist=None, group=None, lanes=2, turns=True, depth=1):
        """lanes: HIP streams; depth: Aligners (contexts, buffer sets) per stream, taken in turn — with depth 2 a
        stream's next block is enqueued while its previous one is still unsettled (the device never waits for the
        host's settle of two steps ago), and still only ONE other chain runs beside an extraction kernel."""
        import torch

        self.torch = torch
        self.device = torch.device("cuda", device)
        base = [torch.cuda.Stream(self.device) for _ in range(lanes)]
        self.streams = [base[i % lanes] for i in range(lanes * depth)]  # (aligner i runs on stream i % lanes)
        self.nstreams, self.depth = lanes, depth
        self.lanes = []
        self.groups = lane_groups(dist, group, lanes * depth)
        for s, g in zip(self.streams, self.groups):
            with torch.cuda.stream(s):  # (a Context adopts the stream that is current when it is made)
                self.lanes.append(Aligner(width, height, kmax=kmax, device=device, dist=dist, group=g))
        # turns=False leaves the lanes' extraction kernels free-running: they overlap head to tail and the step gets
        # ~4 % shorter (0.462 against 0.480 ms), but two of them then share the chip for most of their time and a
        # launch's duration says nothing about the kernel any more — bench.py times the headline with turns and
        # reports the free-running rate beside it
        self.set_turns(turns)
        self._turn = 0
        self._pending = [None] * (lanes * depth)

    def set_turns(self, on):
        """Extraction kernels of the lanes one after the other (rmx_kpe_take_turns) or free-running."""
        if len(self.lanes) > 1:
            if on:
                L.kpe_take_turns([a.ctx for a in self.lanes])
            else:
                for a in self.lanes:
                    L.kpe_take_turns([a.ctx])  # (a group of one: leaves the group)

    def align(self, frames, n_total=None):
        t = self.torch
        i = self._turn % len(self.lanes)
        self._turn += 1
        s = self.streams[i]
        s.wait_stream(t.cuda.current_stream(self.device))  # the frames were produced there
        frames.record_stream(s)  # ... and are read on this lane's stream: not to be recycled under it

        with t.cuda.stream(s):
            return self.lanes[i].align(frames, n_total=n_total, defer=True)

    def settle(self, out):
        """Wait for the block that returned `out` and complete it (status words; a block that has tickets left
        for rmx_kpm_resolve, or that needed a matcher instance outside its lane's launch set, is redone and
        scanned again): its offsets and positions are final after this — or once its lane has taken its next
        block, or after finish().  Returns its fragment count."""
        for lane, s in zip(self.lanes, self.streams):
            if lane._deferred is not None and lane._deferred[0] is out:
                with self.torch.cuda.stream(s):
                    r = lane._settle()
                self.torch.cuda.current_stream(self.device).wait_stream(s)
                return r
        return out["nfrag"]  # (settled already: by its lane's next block or by finish())

    def finish(self):
        t = self.torch
        used = min(self._turn, len(self.lanes))
        last = (self._turn - 1) % len(self.lanes) if self._turn else 0
        nfrag, err = None, None
        for i in range(used):  # every lane is joined before an error is raised
            try:
                with t.cuda.stream(self.streams[i]):
                    r = self.lanes[i].finish()
                if i == last:
                    nfrag = r
            except L.RemapError as e:
                err = err or e
            t.cuda.current_stream(self.device).wait_stream(self.streams[i])
        if err is not None:
            raise err
        return nfrag

    # ---- blocks with host-side stages (canvas blit, codec): submit / drain -------------------------
    def submit(self, frames, n_total=None, blit=False, compress=False):
        """Enqueue the alignment of one block on the next lane; the block that lane held before is
        completed first (status check, fragment count and — if asked for when IT was submitted —
        codec and canvas blit, whose extents need the positions on the host) and returned, else
        None.  The host waits for a block that was enqueued `lanes` submits ago while the other
        lanes keep the device busy; the blit of block t runs beside the extraction of block t + 1."""
        i = self._turn % len(self.lanes)
        done = self._complete(i)
        out = self.align(frames, n_total=n_total)
        self._pending[i] = (out, frames, blit, compress)
        return done

    def drain(self):
        """Complete what is still in flight, oldest first; returns those blocks' results."""
        res = []
        k = len(self.lanes)
        for j in range(k):
            r = self._complete((self._turn + j) % k)
            if r is not None:
                res.append(r)
        for s in self.streams:
            self.torch.cuda.current_stream(self.device).wait_stream(s)
        return res

    def _complete(self, i):
        pend = self._pending[i]
        if pend is None:
            return None
        self._pending[i] = None
        out, frames, blit, compress = pend
        lane = self.lanes[i]
        t = self.torch
        with t.cuda.stream(self.streams[i]):
            out["nfrag"] = lane.finish()
            n = frames.shape[0]
            if compress:
                out["nic_image"] = lane.ctx.nic_compress(frames.reshape(n, -1), pack=False)
                out["nic_median"] = lane.ctx.nic_compress(out["median"].reshape(n, -1), pack=False)
                lane.ctx._chk(lane.ctx.lib.rmx_nic_status(lane.ctx.h), "rmx_nic_status")
            if blit:
                lane.gather_positions(out)
                # the blit goes beside the other lane's small kernels, not beside its extraction
                # kernel (together those two take the sum of their times): it waits for the
                # extraction that is under way, this lane's next one follows it in stream order
                # (the wait goes right before the blit's kernels: the host-side stage — positions to
                # the host, extents — runs while that extraction kernel does)
                lane.blit_gate = lane.ctx.kpe_wait_turn if len(self.lanes) > 1 else None
                out["canvases"] = lane.blit(frames, out["pos_all"], out["frag_all"], out["lo"], out["hi"])
                lane.blit_gate = None
        return out

    def close(self):
        for a in self.lanes:
            a.close()
