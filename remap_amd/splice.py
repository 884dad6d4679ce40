"""Fragment splicing — the host side of fgs::splice (reference fgs.hpp:80-213) on top of the
C ABI.  Everything that touches pixels or keypoints runs on the device:
  extract_single  fgm blend (rmx_fgm_blend) + kpe with grid<1,1>, overlap 0 (rmx_kpe_batch)
  match_*         cellular kpm::match for fragment pairs (rmx_kpm_cell_pairs)
  splice_single   canvas growth + merge (rmx_fgm_merge)
The greedy merge loop, its edge lists and their ORDER (select_match takes the first maximum,
fgs.hpp:142-163) are inherently sequential and stay on the host, mirrored step by step.
"""
from __future__ import annotations

import numpy as np

from . import lib as L

CELL = (15, 15)  # fgs.hpp:121
# Rows of a snippet's descriptor table: extraction starts with SNIPPET_KMAX and is repeated with
# the next power of two when a canvas has more keypoints (the reference has no limit), up to
# SNIPPET_KMAX_LIMIT = the largest table a context takes.  rmx_kpm_cell_pairs matches pairs of
# any size; CELL_CTX_KMAX only sets the size of the tables it matches in LDS.
SNIPPET_KMAX = 2048
SNIPPET_KMAX_LIMIT = 1 << 22
CELL_CTX_KMAX = 4096


class Fragment:
    """fgm::fragment with its canvas in HBM: (ch, cw, 16) int16 tensor, zero, step, frames."""

    def __init__(self, canvas, zero, step, frames):
        self.canvas, self.zero, self.step = canvas, (int(zero[0]), int(zero[1])), step
        self.frames = [(int(n), (int(p[0]), int(p[1]))) for n, p in frames]

    @property
    def dims(self):
        return self.canvas.shape[1], self.canvas.shape[0]

    def normalize(self):  # fgm.hpp:137-143
        zx, zy = self.zero
        self.frames = [(n, (x - zx, y - zy)) for n, (x, y) in self.frames]
        self.zero = (0, 0)

    def _ensure(self, ctx, pos, dim):  # fgm.hpp:190-233
        import torch

        (zx, zy), (w, h), (sw, sh) = self.zero, self.dims, self.step
        up = lambda c, s: (c + s - 1) // s * s
        l = up(zx - pos[0], sw) if pos[0] < zx else 0
        t = up(zy - pos[1], sh) if pos[1] < zy else 0
        rx, ry = pos[0] + dim[0], pos[1] + dim[1]
        r = up(rx - (zx + w), sw) if rx > 0 and rx > zx + w else 0
        b = up(ry - (zy + h), sh) if ry > 0 and ry > zy + h else 0
        if l or t or r or b:
            big = torch.zeros((h + t + b, w + l + r, 16), dtype=torch.int16, device=self.canvas.device)
            ctx.fgm_merge(big, self.canvas, (l, t))
            self.canvas = big
            self.zero = (zx - l, zy - t)

    def blit_fragment(self, ctx, pos, other):  # fgm.hpp:99-113
        self._ensure(ctx, pos, other.dims)
        ctx.fgm_merge(self.canvas, other.canvas, (pos[0] - self.zero[0], pos[1] - self.zero[1]))
        ox, oy = other.zero
        self.frames += [(n, (x - ox + pos[0], y - oy + pos[1])) for n, (x, y) in other.frames]


def shard_pairs(npairs, rank, world):
    """Block-cyclic share of a pair list: indices rank, rank + world, ..."""
    return list(range(rank, npairs, world))


def score_pairs_sharded(score, npairs, dist=None, group=None):
    """Cellular kpm::match votes for `npairs` fragment pairs, the pair list cut block-cyclically
    over the ranks of `group`: every rank scores its share with `score(indices) -> VOTE_DTYPE
    array` (one rmx_kpm_cell_pairs call) and ONE all-gather of the fixed-size vote records
    (16 B per pair, shares padded to equal length) gives every rank the whole vote vector, in
    pair order.  Without torch.distributed: score(all)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size(group) == 1:
        return score(list(range(npairs)))
    import torch

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    mine = shard_pairs(npairs, rank, world)
    share = (npairs + world - 1) // world
    buf = np.zeros(share, L.VOTE_DTYPE)
    if mine:
        buf[: len(mine)] = score(mine)
    flat = torch.from_numpy(buf.view(np.int32).reshape(share, 4).copy())
    backend = dist.get_backend(group)
    if backend != "gloo":
        flat = flat.cuda()
    out = torch.empty((world * share, 4), dtype=torch.int32, device=flat.device)
    dist.all_gather_into_tensor(out, flat, group=group)
    allv = out.cpu().numpy().reshape(world, share, 4)
    votes = np.zeros(npairs, L.VOTE_DTYPE)
    for r in range(world):
        idx = shard_pairs(npairs, r, world)
        votes[idx] = np.ascontiguousarray(allv[r, : len(idx)]).view(L.VOTE_DTYPE).reshape(-1)
    return votes


class _Snippet:
    def __init__(self, fragment, mask, table, k):
        self.fragment, self.mask, self.table, self.k = fragment, mask, table, k
        self.edges = []  # [primary, (dx, dy), count, other snippet]


def _extract_single(ctx_cache, frag, device):  # fgs.hpp:80-89
    any_ctx = ctx_cache("any")
    img, mask = any_ctx.fgm_blend(frag.canvas)
    w, h = frag.dims
    kmax = SNIPPET_KMAX
    while True:
        kctx = ctx_cache((w, h, kmax))
        _, desc, kc = kctx.kpe(img[None].contiguous())
        rc, maxk = kctx.kpe_status()
        if rc == L.RMX_ERR_KMAX and maxk <= SNIPPET_KMAX_LIMIT:  # same canvas, larger table
            while kmax < maxk:
                kmax *= 2
            continue
        if rc:
            raise L.RemapError(rc, "rmx_kpe_status",
                               f"a fragment of {w}x{h} has {maxk} keypoints; a descriptor table holds at most "
                               f"{SNIPPET_KMAX_LIMIT}")
        return _Snippet(frag, mask, desc[0], int(kc[0].item()))


def _extract_all(ctx_cache, frags):  # fgs.hpp:91-103
    """extract_single for every fragment, batched: the blends are enqueued one after the other (no host
    wait between them), fragments of equal canvas size share ONE rmx_kpe_batch launch and ONE status /
    count read-back (the reference runs them under std::execution::par; a stream with hundreds of
    fragments otherwise pays a device round trip per fragment before the first pair is scored)."""
    import torch

    any_ctx = ctx_cache("any")
    blends = [any_ctx.fgm_blend(f.canvas) for f in frags]
    groups = {}
    for i, f in enumerate(frags):
        groups.setdefault(f.dims, []).append(i)
    snippets = [None] * len(frags)
    for (w, h), idx in groups.items():
        imgs = torch.stack([blends[i][0] for i in idx])
        kmax = SNIPPET_KMAX
        while True:
            kctx = ctx_cache((w, h, kmax))
            _, desc, kc = kctx.kpe(imgs)
            rc, maxk = kctx.kpe_status()
            if rc == L.RMX_ERR_KMAX and maxk <= SNIPPET_KMAX_LIMIT:  # same canvases, larger tables
                while kmax < maxk:
                    kmax *= 2
                continue
            if rc:
                raise L.RemapError(rc, "rmx_kpe_status",
                                   f"a fragment of {w}x{h} has {maxk} keypoints; a descriptor table holds at most "
                                   f"{SNIPPET_KMAX_LIMIT}")
            break
        ks = kc.cpu().numpy()
        for j, i in enumerate(idx):
            snippets[i] = _Snippet(frags[i], blends[i][1], desc[j], int(ks[j]))
    return snippets


def splice(fragments, device=0, dist=None, group=None):
    """fgs::splice over a list of normalised Fragments (collector.complete()).  Returns the
    spliced fragments in the reference's list order.

    With torch.distributed (`dist`), every rank holds the same fragments and runs the same
    greedy loop; only the pair scoring — the O(F^2) part — is cut over the ranks
    (score_pairs_sharded), so all ranks see identical votes and stay in step without any
    other exchange."""
    ctxs = {}

    def ctx_cache(key):
        if key not in ctxs:
            w, h, kmax = (16, 8, CELL_CTX_KMAX) if key == "any" else key
            ctxs[key] = L.Context(w, h, grid=(1, 1), overlap=0, kmax=kmax, device=device)
        return ctxs[key]

    mctx = ctx_cache("any")
    snippets = _extract_all(ctx_cache, fragments)

    def score(members, pairs):
        """votes of `pairs` (indices into `members`), scored on this rank's share only"""
        def run(idx):
            return mctx.kpm_cell_pairs([s.table for s in members], [s.k for s in members], [s.mask for s in members],
                                       [s.fragment.dims for s in members], [pairs[i] for i in idx], CELL)
        return score_pairs_sharded(run, len(pairs), dist, group)

    def bind(head, other, v):  # snippet::bind, fgs.hpp:56-65
        if v["valid"]:
            off, cnt = (int(v["dx"]), int(v["dy"])), int(v["count"])
            head.edges.append([True, off, cnt, other])
            other.edges.append([False, (-off[0], -off[1]), cnt, head])

    def match_partial(head, rest):  # fgs.hpp:119-133
        if not rest:
            return
        members = [head] + rest
        for v, other in zip(score(members, [(0, i) for i in range(1, len(members))]), rest):
            bind(head, other, v)

    # match_all (fgs.hpp:135-140): match_partial(i, rest) for every i — one pair list, i-major,
    # so edges are bound in the reference's order
    all_pairs = [(i, j) for i in range(len(snippets)) for j in range(i + 1, len(snippets))]
    if all_pairs:
        for (i, j), v in zip(all_pairs, score(snippets, all_pairs)):
            bind(snippets[i], snippets[j], v)

    while True:
        best = None  # select_match: first maximum over primary edges in list order (fgs.hpp:142-163)
        for s in snippets:
            for e in s.edges:
                if e[0] and (best is None or e[2] > best[1][2]):
                    best = (s, e)
        if best is None:
            break
        left, edge = best
        right = edge[3]
        dst = left.fragment
        pos = (dst.zero[0] + edge[1][0], dst.zero[1] + edge[1][1])
        dst.blit_fragment(mctx, pos, right.fragment)  # splice_single, fgs.hpp:165-183
        dst.normalize()
        merged = _extract_single(ctx_cache, dst, device)
        for s in (right, left):  # unbind: drop the mirrored edges held by the others
            for e in s.edges:
                e[3].edges = [x for x in e[3].edges if x[3] is not s]
        snippets = [merged] + [s for s in snippets if s is not right and s is not left]
        match_partial(merged, snippets[1:])
    out = [s.fragment for s in snippets]
    for c in ctxs.values():
        c.close()
    return out
