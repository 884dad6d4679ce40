"""nic::compress on the GPU (rmx_nic_compress_batch) against the reference's golden streams
and the CPU oracle, through the C ABI."""
import numpy as np
import pytest

import orclib as O
from remap_amd import lib as L
from remap_amd.synth import scroller_v1
from test_oracle_golden import nic_cases

pytestmark = pytest.mark.gpu


def gpu_compress(ctx, imgs):
    """imgs: list of equally sized uint8 arrays -> list of compressed byte strings"""
    import torch

    batch = torch.from_numpy(np.stack([np.ascontiguousarray(i).reshape(-1) for i in imgs])).cuda()
    packed, sizes = ctx.nic_compress(batch)
    pk = packed.cpu().numpy().tobytes()
    offs = np.concatenate([[0], np.cumsum(sizes)])
    return [pk[offs[i]:offs[i + 1]] for i in range(len(imgs))]


def test_nic_golden_cases(gpu):
    ctx = L.Context(16, 8, grid=(1, 1), overlap=0, kmax=64)
    by_size = {}
    for k, img, pk in nic_cases():
        by_size.setdefault(img.size, []).append((k, img, pk))
    for size, group in by_size.items():  # one batch per image size
        got = gpu_compress(ctx, [g[1] for g in group])
        for (k, img, pk), g in zip(group, got):
            assert g == pk, (k, img.shape, len(g), len(pk))
    ctx.close()


def test_nic_frames_and_medians_match_oracle(gpu):
    import torch

    fr = np.concatenate([scroller_v1(24, dens=200)[0], scroller_v1(8, dens=40)[0]])
    ctx = L.Context(320, 240, kmax=2048)
    d = torch.from_numpy(fr).cuda()
    med, _, _ = ctx.kpe(d)
    for plane in (d, med):
        got = gpu_compress(ctx, list(plane.cpu().numpy()))
        for i, g in enumerate(got):
            assert g == O.orc_nic_compress(plane[i].cpu().numpy()), i
    ctx.close()


def test_nic_beyond_reference_domain(gpu):
    # a blank frame (one run of 76800 > 65535), pure noise (literal of 76800 > 16383), runs of
    # 65536..65538 and 2 x 65535 +- 1: split tokens, equal to the oracle, and they decode
    rng = np.random.default_rng(8)
    P = 320 * 240
    imgs = [np.full(P, 6, np.uint8), rng.integers(0, 16, size=P, dtype=np.uint8)]
    for r in (65536, 65537, 65538, 76797):
        x = rng.integers(0, 16, size=P, dtype=np.uint8)
        x[2:2 + r] = 9
        imgs.append(x)
    x = np.zeros(P, np.uint8)
    x[::2] = 1  # literal with no run at all
    imgs.append(x)
    ctx = L.Context(320, 240, kmax=64)
    got = gpu_compress(ctx, imgs)
    for i, (img, g) in enumerate(zip(imgs, got)):
        assert g == O.orc_nic_compress(img), i
        dec, cnt = O.orc_nic_decompress(g, P)
        assert cnt == P and np.array_equal(dec, img), i
    ctx.close()


def test_nic_stride_too_small_is_reported(gpu):
    import ctypes as C

    import torch

    ctx = L.Context(16, 8, kmax=64, grid=(1, 1), overlap=0)
    img = torch.from_numpy(np.random.default_rng(1).integers(0, 16, size=(2, 4096), dtype=np.uint8)).cuda()
    blobs = torch.zeros((2, 64), dtype=torch.uint8, device="cuda")
    sizes = torch.zeros(2, dtype=torch.int32, device="cuda")
    rc = ctx.lib.rmx_nic_compress_batch(ctx.h, img.data_ptr(), 2, 4096, blobs.data_ptr(), 64, sizes.data_ptr())
    assert rc == 0
    assert ctx.lib.rmx_nic_status(ctx.h) == -4  # RMX_ERR_CAPACITY
    assert int(sizes[0]) > 64  # the true size is still reported
    assert ctx.lib.rmx_nic_status(ctx.h) == 0
    ctx.close()


def test_nic_decompress_golden_streams_and_round_trip(gpu):
    """rmx_nic_decompress_batch decodes the reference's own streams to their images, inverts
    rmx_nic_compress_batch on whole frames, and reports a malformed stream."""
    import torch

    ctx = L.Context(16, 8, grid=(1, 1), overlap=0, kmax=64)
    by_size = {}
    for k, img, pk in nic_cases():
        by_size.setdefault(img.size, []).append((img, pk))
    for size, group in by_size.items():
        packed = torch.from_numpy(np.frombuffer(b"".join(g[1] for g in group), np.uint8).copy()).cuda()
        out = ctx.nic_decompress(packed, [len(g[1]) for g in group], size).cpu().numpy()
        for row, (img, _) in zip(out, group):
            assert np.array_equal(row, img.reshape(-1))
    fr = np.concatenate([scroller_v1(6, dens=200)[0], scroller_v1(3, dens=40)[0],
                         np.full((1, 240, 320), 6, np.uint8),
                         np.random.default_rng(2).integers(0, 16, size=(1, 240, 320), dtype=np.uint8)])
    d = torch.from_numpy(fr).cuda()
    packed, sizes = ctx.nic_compress(d.reshape(len(fr), -1))
    back = ctx.nic_decompress(packed, sizes, 320 * 240)
    assert torch.equal(back, d.reshape(len(fr), -1))
    # truncated stream / wrong pixel count -> RMX_ERR_ARG
    with pytest.raises(L.RemapError):
        ctx.nic_decompress(packed, [int(sizes[0]) - 1], 320 * 240)
    with pytest.raises(L.RemapError):
        ctx.nic_decompress(packed, [int(sizes[0])], 320 * 240 - 5)
    ctx.close()


def _random_stream(rng, kinds):
    """A valid token stream of every header form (nic.hpp:103-164) and the pixels it decodes to."""
    toks, px = [], []
    for _ in range(int(rng.integers(1, 400))):
        kind = kinds[int(rng.integers(len(kinds)))]
        v = int(rng.integers(16))
        if kind == "r1":
            n = int(rng.integers(3, 7))
            toks.append(bytes([(n - 3) << 4 | v]))
        elif kind == "r2":
            n = int(rng.integers(0, 256))
            toks.append(bytes([0x50 | v, n]))
        elif kind == "r3":
            n = int(rng.integers(0, 3000))
            toks.append(bytes([0x60 | v, n & 255, n >> 8]))
        elif kind == "r0":
            n = 0
            toks.append(bytes([0x40 | v]))
        if kind[0] == "r":
            px.append(np.full(n, v, np.uint8))
            continue
        n = int(rng.integers(0, 64)) if kind == "l1" else int(rng.integers(0, 700))
        p = rng.integers(0, 16, size=n, dtype=np.uint8)
        q = np.concatenate([p, np.zeros(n & 1, np.uint8)])
        body = bytes((q[0::2] << 4 | q[1::2]).astype(np.uint8))
        toks.append((bytes([0x80 | n]) if kind == "l1" else bytes([0xc0 | n >> 8, n & 255])) + body)
        px.append(p)
    return b"".join(toks), np.concatenate(px)


def test_nic_decompress_random_token_streams(gpu):
    """Every header form, zero-length tokens, tokens across the 64-byte chunk and 4-KB window
    edges, blobs at odd offsets: the decoder equals the oracle's on each stream."""
    import torch

    rng = np.random.default_rng(77)
    ctx = L.Context(16, 8, grid=(1, 1), overlap=0, kmax=64)
    mixes = [("r1",), ("r1", "l1"), ("r1", "r2", "r3", "r0", "l1", "l2"), ("l1", "l2"), ("r3", "l2"), ("r1", "r1", "r1", "l1", "r2")]
    for rep in range(24):
        streams = [_random_stream(rng, mixes[(rep + j) % len(mixes)]) for j in range(9)]
        for blob, pix in streams:
            dec, cnt = O.orc_nic_decompress(blob, len(pix))
            assert cnt == len(pix) and np.array_equal(dec, pix)
        # one launch decodes images of one pixel count: pad each stream with runs up to the longest
        P = max(len(p) for _, p in streams) + 7
        blobs, want = [], []
        for blob, pix in streams:
            pad = P - len(pix)
            tail = b""
            while pad > 0:
                n = min(pad, 255) if pad >= 3 else pad
                tail += bytes([0x50 | 3, n])
                pad -= n
            blobs.append(blob + tail)
            want.append(np.concatenate([pix, np.full(P - len(pix), 3, np.uint8)]))
        packed = torch.from_numpy(np.frombuffer(b"".join(blobs), np.uint8).copy()).cuda()
        out = ctx.nic_decompress(packed, [len(b) for b in blobs], P).cpu().numpy()
        for j, w in enumerate(want):
            assert np.array_equal(out[j], w), (rep, j)
    ctx.close()
