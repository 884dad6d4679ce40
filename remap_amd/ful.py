"""Fragment files — ful::write / ful::read of the reference (ful.hpp:10-102) for fragments
whose canvas lives in HBM.  One file per fragment, named by its index in the list; bytes
(LP64 little endian, exactly what the reference streams out):

    u64 width, u64 height                      mrl::dimensions_t       (ful.hpp:16-17)
    u16 dots[height][width][16]                fgm::dot_type canvas    (ful.hpp:19-20)
    i32 zero_x, i32 zero_y                     fgm::point_t            (ful.hpp:22-23)
    u64 frame_count                                                    (ful.hpp:25-26)
    per frame: u64 number, i32 x, i32 y, u64 size_i, bytes, u64 size_m, bytes   (ful.hpp:28-43)

read() restores fragments with step (1, 1), as the reference does (ful.hpp:97-98).  The
checkpoint between collect and splice."""
from __future__ import annotations

import os
import struct

import numpy as np

from .splice import Fragment

__all__ = ["write", "read", "pack", "unpack"]


def _frame_fields(fr):
    no, pos = fr[0], fr[1]
    bi = bytes(fr[2]) if len(fr) > 2 and fr[2] is not None else b""
    bm = bytes(fr[3]) if len(fr) > 3 and fr[3] is not None else b""
    return int(no), (int(pos[0]), int(pos[1])), bi, bm


def pack(dots, zero, frames) -> bytes:
    """File image of one fragment.  dots: (h, w, 16) uint16 host array; frames: iterable of
    (number, (x, y)) or (number, (x, y), image_blob, median_blob)."""
    dots = np.ascontiguousarray(dots).view(np.uint16)
    h, w = dots.shape[0], dots.shape[1]
    out = [struct.pack("<QQ", w, h), dots.tobytes(), struct.pack("<ii", int(zero[0]), int(zero[1]))]
    frames = list(frames)
    out.append(struct.pack("<Q", len(frames)))
    for fr in frames:
        no, (x, y), bi, bm = _frame_fields(fr)
        out += [struct.pack("<Qii", no, x, y), struct.pack("<Q", len(bi)), bi, struct.pack("<Q", len(bm)), bm]
    return b"".join(out)


def unpack(buf: bytes):
    """Inverse of pack: (dots (h, w, 16) uint16, zero, [(number, (x, y), image_blob, median_blob)])."""
    w, h = struct.unpack_from("<QQ", buf, 0)
    o = 16
    n = w * h * 16
    dots = np.frombuffer(buf, np.uint16, n, o).reshape(h, w, 16).copy()
    o += 2 * n
    zx, zy = struct.unpack_from("<ii", buf, o)
    o += 8
    (cnt,) = struct.unpack_from("<Q", buf, o)
    o += 8
    frames = []
    for _ in range(cnt):
        no, x, y = struct.unpack_from("<Qii", buf, o)
        o += 16
        (ni,) = struct.unpack_from("<Q", buf, o)
        o += 8
        bi = bytes(buf[o:o + ni])
        o += ni
        (nm,) = struct.unpack_from("<Q", buf, o)
        o += 8
        bm = bytes(buf[o:o + nm])
        o += nm
        frames.append((no, (x, y), bi, bm))
    if o != len(buf):
        raise ValueError(f"fragment file: {len(buf) - o} trailing bytes")
    return dots, (zx, zy), frames


def write(directory, fragments, blobs=None):
    """ful::write(dir, first, last): fragment i -> directory/str(i).  `blobs` optionally maps a
    frame number to its (image_blob, median_blob) (the collector's compressed frames)."""
    os.makedirs(directory, exist_ok=True)
    for i, f in enumerate(fragments):
        dots = f.canvas.cpu().numpy().view(np.uint16)
        frames = [(no, pos) + tuple(blobs.get(no, (b"", b""))) if blobs else (no, pos) for no, pos in f.frames]
        with open(os.path.join(directory, str(i)), "wb") as fh:
            fh.write(pack(dots, f.zero, frames))


def read(directory, device=0, with_blobs=False):
    """ful::read(dir): files sorted by their integer names.  Canvases are uploaded to `device`.
    Returns the fragments (and, if asked, {frame number: (image_blob, median_blob)})."""
    import torch

    names = sorted(os.listdir(directory), key=int)
    frags, blobs = [], {}
    for name in names:
        with open(os.path.join(directory, name), "rb") as fh:
            dots, zero, frames = unpack(fh.read())
        canvas = torch.from_numpy(dots.view(np.int16)).to(torch.device("cuda", device))
        frags.append(Fragment(canvas, zero, (1, 1), [(no, pos) for no, pos, _, _ in frames]))
        for no, _, bi, bm in frames:
            blobs[no] = (bi, bm)
    return (frags, blobs) if with_blobs else frags
