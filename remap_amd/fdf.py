"""Foreground filtering — the host side of fdf::filter (reference fdf.hpp:40-89) on top of the
C ABI: per fragment, the background is the blend of its canvas (fdf.hpp:22-36) and every frame
of the fragment goes through contour extraction, masking and the masked blit on the device
(rmx_fdf_filter_batch).  Frames and medians are taken from device tensors indexed by frame
number (resident after collect, or decoded from the stored nic blobs with
Context.nic_decompress)."""
from __future__ import annotations

import numpy as np

from . import lib as L
from .splice import Fragment

__all__ = ["filter_fragments"]


def filter_fragments(fragments, frames, medians, width, height, device=0, ctx=None, want_masks=False):
    """fdf::filter(fragments, frame_dim, comp, cb).

    fragments: list of splice.Fragment (canvas on the device; frames = [(number, (x, y))]);
    frames / medians: (N, H, W) uint8 CUDA tensors holding image and median of frame `number`
    at index `number`.  Returns the filtered fragments (same dimensions and zero as their
    background, frames re-blitted with the foreground masked out, fdf.hpp:55-72) and, if
    want_masks, {frame number: (mask (H, W) tensor, kept contours)}."""
    import torch

    own = ctx is None
    if own:
        ctx = L.Context(width, height, kmax=64, device=device)
    out, info = [], {}
    for frag in fragments:
        bg, _ = ctx.fgm_blend(frag.canvas)  # details::get_background
        nos = torch.tensor([no for no, _ in frag.frames], dtype=torch.long, device=frames.device)
        zx, zy = frag.zero
        pos = torch.tensor([[x - zx, y - zy] for _, (x, y) in frag.frames], dtype=torch.int32, device=frames.device)
        result = torch.zeros_like(frag.canvas)
        if len(frag.frames):
            fr = frames.index_select(0, nos).contiguous()
            md = medians.index_select(0, nos).contiguous()
            masks, ncont = ctx.fdf_filter(bg.contiguous(), fr, md, pos.contiguous(), canvas=result)
            if want_masks:
                nc = ncont.cpu().numpy()
                for k, (no, _) in enumerate(frag.frames):
                    info[no] = (masks[k], int(nc[k]))
        out.append(Fragment(result, frag.zero, (1, 1), list(frag.frames)))
    if own:
        ctx.close()
    return (out, info) if want_masks else out
