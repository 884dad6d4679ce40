"""Synthetic frame streams for tests and bench.

The reference ships no sample data (/data is git-ignored, .gitignore:6), so every workload
is generated.  ``scroller_v1`` is the generator the survey verified against the reference:
a side-scroller world of random rectangles and a camera that moves right by 0..4 px per
frame.  ``std::mt19937`` streams are reproduced with numpy's legacy ``RandomState`` whose
``init_genrand`` seeding and raw 32-bit outputs are identical (checked in
tests/test_synth.py against libstdc++).
"""
from __future__ import annotations

import numpy as np

__all__ = ["mt19937_raw", "scroller_v1", "scroller_world", "rooms_v1", "upscale2"]


def mt19937_raw(seed: int, n: int) -> np.ndarray:
    """First ``n`` outputs of ``std::mt19937(seed)`` as uint32."""
    return _raw(np.random.RandomState(seed), n)


def _raw(rs: np.random.RandomState, n: int) -> np.ndarray:
    # randint over the full 32-bit range returns the raw tempered words unmodified.
    return rs.randint(0, 0x1_0000_0000, size=n, dtype=np.uint32)


def scroller_world(n_frames: int, width: int = 320, height: int = 240, dens: int = 200,
                   world_seed: int = 42, cam_seed: int = 1234, max_step: int = 5):
    """World canvas + camera x positions of scroller-v1.

    world: (height, Ww) uint8 of palette indices 0..15; cam_x: (n_frames,) int64.
    Ground truth offsets: (dx, dy) = (cam_x[i] - cam_x[i-1], 0).
    """
    steps = _raw(np.random.RandomState(cam_seed), n_frames) % max_step
    cam_x = np.zeros(n_frames, dtype=np.int64)
    cam_x[1:] = np.cumsum(steps[: n_frames - 1].astype(np.int64))
    # all N draws are made before sizing the world (x_N is the position after the last draw)
    x_n = int(cam_x[-1] + int(steps[n_frames - 1])) if n_frames else 0
    ww = x_n + width + 8
    world = np.full((height, ww), 6, dtype=np.uint8)
    n_rect = ww * height // dens
    r = _raw(np.random.RandomState(world_seed), 5 * n_rect).astype(np.int64)
    x0 = r[0::5] % ww
    y0 = r[1::5] % height
    rw = 1 + r[2::5] % 12
    rh = 1 + r[3::5] % 12
    col = (r[4::5] % 16).astype(np.uint8)
    for i in range(n_rect):
        world[y0[i]: y0[i] + rh[i], x0[i]: x0[i] + rw[i]] = col[i]
    return world, cam_x


def scroller_v1(n_frames: int, width: int = 320, height: int = 240, dens: int = 200, **kw):
    """(frames, truth): frames (n, height, width) uint8 C-contiguous; truth (n, 2) int32 with
    truth[0] = (0, 0) and truth[i] = (dx, dy) of frame i relative to frame i-1."""
    world, cam_x = scroller_world(n_frames, width, height, dens, **kw)
    win = np.lib.stride_tricks.sliding_window_view(world, width, axis=1)  # (H, Ww-W+1, W)
    frames = np.ascontiguousarray(win[:, cam_x, :].transpose(1, 0, 2))
    truth = np.zeros((n_frames, 2), dtype=np.int32)
    truth[1:, 0] = np.diff(cam_x)
    return frames, truth


def upscale2(frames: np.ndarray) -> np.ndarray:
    """Pixel-double every frame (config 4: 640x480 'upscaled capture')."""
    return np.ascontiguousarray(frames.repeat(2, axis=1).repeat(2, axis=2))


def rooms_v1(n_frames: int, width: int = 320, height: int = 240, n_rooms: int = 4, dens: int = 200,
             seed: int = 7):
    """Top-down multi-room stream (config 5): a 2-D camera (dx,dy in -2..2) wanders inside
    disjoint room rectangles; a blank single-colour frame separates rooms (forcing a
    fragment break) and every room is visited twice so spliceable overlaps exist.

    Returns (frames, room_id, cam_xy); blank frames have room_id = -1.
    """
    visits = 2 * n_rooms
    per = max(2, (n_frames - visits) // visits)
    rw, rh = width * 3, height * 3
    rooms = []
    for k in range(n_rooms):
        w = np.full((rh, rw), 6, dtype=np.uint8)
        n_rect = rw * rh // dens
        r = _raw(np.random.RandomState(seed * 1000 + k), 5 * n_rect).astype(np.int64)
        for i in range(n_rect):
            x0, y0 = r[5 * i] % rw, r[5 * i + 1] % rh
            w[y0: y0 + 1 + r[5 * i + 3] % 12, x0: x0 + 1 + r[5 * i + 2] % 12] = r[5 * i + 4] % 16
        rooms.append(w)
    cam = _raw(np.random.RandomState(seed), 2 * n_frames + 4 * visits).astype(np.int64)
    frames = np.empty((n_frames, height, width), dtype=np.uint8)
    room_id = np.full(n_frames, -1, dtype=np.int32)
    cam_xy = np.zeros((n_frames, 2), dtype=np.int32)
    f = 0
    ci = 0
    for v in range(visits):
        k = v % n_rooms
        x = int(cam[ci] % (rw - width)); y = int(cam[ci + 1] % (rh - height)); ci += 2
        if v >= n_rooms:  # second visit starts near the first visit's track
            x, y = min(rw - width, width // 2 + 10 * k), min(rh - height, height // 2)
        for _ in range(per):
            if f >= n_frames:
                break
            frames[f] = rooms[k][y: y + height, x: x + width]
            room_id[f] = k
            cam_xy[f] = (x, y)
            f += 1
            x = int(np.clip(x + int(cam[ci] % 5) - 2, 0, rw - width))
            y = int(np.clip(y + int(cam[ci + 1] % 5) - 2, 0, rh - height)); ci += 2
        if f < n_frames:
            frames[f] = 6
            f += 1
    while f < n_frames:
        frames[f] = 6
        f += 1
    return frames, room_id, cam_xy


def with_sprites(frames, n_sprites=2, size=9, seed=5):
    """Overlay small multi-coloured sprites that move on the SCREEN (not with the world): the
    foreground fdf::filter is there to remove.  Sprite k follows a Lissajous path; its pattern
    is a fixed random size x size block with a few transparent pixels."""
    rng = np.random.default_rng(seed)
    out = frames.copy()
    n, H, W = frames.shape
    pats = [rng.integers(0, 16, size=(size, size), dtype=np.uint8) for _ in range(n_sprites)]
    holes = [rng.random((size, size)) < 0.15 for _ in range(n_sprites)]
    for i in range(n):
        for k in range(n_sprites):
            cx = int((W - size - 8) * (0.5 + 0.45 * np.sin(0.07 * i * (k + 1) + k))) + 4
            cy = int((H - size - 8) * (0.5 + 0.45 * np.cos(0.05 * i * (k + 2) + 2 * k))) + 4
            blk = out[i, cy:cy + size, cx:cx + size]
            blk[~holes[k]] = pats[k][~holes[k]]
    return out
