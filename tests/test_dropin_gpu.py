"""The C++ drop-in header surface (include/remap/*, reference names: frc::collector,
kpe::extractor, kpm::match, fgm::fragment) driven by a caller written like main.cpp's,
on the GPU, against the CPU oracle."""
import os
import subprocess

import numpy as np
import pytest

import orclib as O
from remap_amd import lib as L
from remap_amd.synth import scroller_v1
from test_oracle_golden import poly_sum

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_driver(tmp):
    L.load()
    exe = os.path.join(tmp, "drop_in")
    cmd = ["g++", "-std=c++20", "-O1", "-I" + os.path.join(ROOT, "include"),
           "-I" + os.path.join(ROOT, "include", "remap", "compat"), os.path.join(ROOT, "tests", "cpp", "drop_in.cpp"),
           "-L" + os.path.join(ROOT, "remap_amd"), "-lremap_amd", "-L/opt/rocm/lib", "-lamdhip64",
           "-Wl,-rpath," + os.path.join(ROOT, "remap_amd"), "-Wl,-rpath,/opt/rocm/lib", "-o", exe]
    subprocess.check_call(cmd)
    return exe


def test_collector_drop_in_matches_oracle(gpu, tmp_path):
    exe = build_driver(str(tmp_path))
    frames, truth = scroller_v1(100, dens=100)
    frames = frames.copy()
    frames[40] = 6  # a fragment break
    check_collect(exe, frames, tmp_path)


def test_collector_blocks_with_prefetched_feed(gpu, tmp_path):
    """The same stream in blocks of 7 frames (halo tables, scan carry and fragment ids across 15 block boundaries),
    once with the feed drained on the calling thread and once by the helper thread (prefetch_feed): callbacks in
    order, medians (downloaded when the callback reads them), grids, canvases and positions as the oracle's."""
    exe = build_driver(str(tmp_path))
    frames, truth = scroller_v1(100, dens=100)
    frames = frames.copy()
    frames[40] = 6  # a fragment break
    check_collect(exe, frames, tmp_path, env={"REMAP_TEST_BLOCK": "7"})
    check_collect(exe, frames, tmp_path, env={"REMAP_TEST_BLOCK": "7", "REMAP_TEST_PREFETCH": "1"})


def noisy(p, seed):
    rng = np.random.default_rng(seed)
    img = np.full((240, 320), 6, np.uint8)
    m = rng.random(img.shape) < p
    img[m] = rng.integers(0, 16, int(m.sum()))
    return img


def test_collector_grows_tables_beyond_kmax(gpu, tmp_path):
    """Frames with more keypoints than the collector's initial table stride (2 048 rows): the
    block is run again with larger tables, results equal the oracle's (the reference has no
    limit).  Two such frames follow each other, one a shifted copy of the other, so the dense
    matcher tiers see them too; the second block of the stream starts with one (halo at the
    new stride)."""
    exe = build_driver(str(tmp_path))
    frames, _ = scroller_v1(40, dens=100)
    frames = frames.copy()
    big = noisy(0.25, 5)
    assert O.orc_kpe(big)[2] > 2048
    frames[12] = big
    frames[13] = np.roll(big, -2, axis=1)
    frames[14] = np.roll(big, -3, axis=1)
    check_collect(exe, frames, tmp_path)


def test_collector_resolves_huge_regions(gpu, tmp_path):
    """Frames with a box of pure noise: ~6 000 keypoints, all in one grid region — more than the
    4 096 entries the LDS matcher holds per region.  The collector grows its tables and lets
    rmx_kpm_resolve compute the tickets; callbacks, fragments and canvases equal the oracle's."""
    exe = build_driver(str(tmp_path))
    frames, _ = scroller_v1(16, dens=100)
    frames = frames.copy()
    rng = np.random.default_rng(21)
    big = np.full((240, 320), 6, np.uint8)
    big[6:126, 6:86] = rng.integers(0, 16, (120, 80))
    _, dsc, k = O.orc_kpe(big)
    assert 2048 < k <= 16384 and int((dsc["region_mask"] & 1).sum()) > 4096
    frames[6] = big
    frames[7] = np.roll(big, -2, axis=1)
    frames[8] = np.roll(big, -4, axis=1)
    check_collect(exe, frames, tmp_path)


def test_collector_takes_frames_of_pure_noise(gpu, tmp_path):
    """~46 000 keypoints per frame: the collector grows its tables to 65 536 rows, the matcher
    hands such frames to its global-memory instance — collect never refuses a frame."""
    exe = build_driver(str(tmp_path))
    frames, _ = scroller_v1(10, dens=100)
    frames = frames.copy()
    rng = np.random.default_rng(33)
    noise = rng.integers(0, 16, size=(240, 320), dtype=np.uint8)
    frames[4] = noise
    frames[5] = np.roll(noise, -3, axis=1)
    check_collect(exe, frames, tmp_path)


def check_collect(exe, frames, tmp_path, env=None):
    blob = tmp_path / "frames.bin"
    frames.tofile(blob)
    out = subprocess.run([exe, str(blob), "320", "240"], check=True, capture_output=True, text=True,
                         env=dict(os.environ, **(env or {}))).stdout
    lines = out.strip().splitlines()
    head = dict(zip(lines[0].split()[::2], lines[0].split()[1::2]))
    o = O.orc_frc(frames)
    n = len(frames)
    nfrag = int(o["frag"].max()) + 1
    assert int(head["calls"]) == n - 1 and int(head["ordered"]) == 1 and int(head["nfrag"]) == nfrag
    ent = 0
    medsum = 0
    for i in range(1, n):
        med, desc, k = O.orc_kpe(frames[i])
        ent += sum(bin(int(m)).count("1") for m in desc["region_mask"])
        medsum += int(med.sum())
    assert int(head["entries"]) == ent and int(head["medsum"]) == medsum
    frag_lines = [l.split() for l in lines if l.startswith("fragment ")]
    frame_lines = [l.split() for l in lines if l.startswith("frame ")]
    assert len(frame_lines) == n
    for f in range(nfrag):
        sel = o["frag"] == f
        cw, ch, zx, zy = O.orc_fgm_extent(o["pos"][sel], 320, 240)
        fl = frag_lines[f]
        assert [int(fl[3]), int(fl[4])] == [cw, ch] and [int(fl[6]), int(fl[7])] == [0, 0]
        assert int(fl[9]) == int(sel.sum())
        can = np.zeros((ch, cw, 16), np.uint16)
        for i in np.nonzero(sel)[0]:
            O.orc_fgm_blit(can, (zx, zy), frames[i], o["pos"][i])
        assert int(fl[11]) == poly_sum(can)
        assert tuple(int(v) for v in fl[13:17]) == O.orc_fgm_margins(can)  # fragment::margins (fgm.hpp:145-153)
    for fl in frame_lines:
        i, f = int(fl[1]), int(fl[3])
        assert f == o["frag"][i]
        sel = o["frag"] == f
        _, _, zx, zy = O.orc_fgm_extent(o["pos"][sel], 320, 240)
        assert [int(fl[5]), int(fl[6])] == (o["pos"][i] - [zx, zy]).tolist()
        assert fl[8] == "1" and fl[9] == "1"  # the caller's compressor ran on image and median
    pair = [l.split() for l in lines if l.startswith("pair ")][0]
    m, d, _ = O.orc_kpm_grid(O.orc_kpe(frames[0])[1], O.orc_kpe(frames[1])[1])
    d0 = O.orc_kpe(frames[0])[1]
    assert int(pair[2]) == sum(bin(int(x)).count("1") for x in d0["region_mask"])
    assert int(pair[4]) == m and (int(pair[6]), int(pair[7])) == d


def oracle_fragments(frames):
    """canvases (poly_sum), dims and frame counts of the oracle's collect over `frames`"""
    o = O.orc_frc(frames)
    res = []
    for f in range(int(o["frag"].max()) + 1):
        sel = o["frag"] == f
        cw, ch, zx, zy = O.orc_fgm_extent(o["pos"][sel], 320, 240)
        can = np.zeros((ch, cw, 16), np.uint16)
        for i in np.nonzero(sel)[0]:
            O.orc_fgm_blit(can, (zx, zy), frames[i], o["pos"][i])
        res.append((cw, ch, int(sel.sum()), poly_sum(can)))
    return o, res


def test_second_collect_opens_a_new_fragment(gpu, tmp_path):
    """collect() twice on one collector: process_init opens a fragment both times
    (frc.hpp:83-95); the second stream must not land in the first one's canvas."""
    exe = build_driver(str(tmp_path))
    frames, _ = scroller_v1(20, dens=100)  # one fragment per pass
    blob = tmp_path / "frames.bin"
    frames.tofile(blob)
    out = subprocess.run([exe, str(blob), "320", "240", "twice"], check=True, capture_output=True, text=True).stdout
    lines = [l.split() for l in out.strip().splitlines()]
    _, want = oracle_fragments(frames)
    assert len(want) == 1
    assert int(lines[0][1]) == 2 * (len(frames) - 1) and int(lines[0][3]) == 2
    for fl in lines[1:]:
        cw, ch, nf, s = want[0]
        assert [int(fl[3]), int(fl[4]), int(fl[9]), int(fl[11])] == [cw, ch, nf, s]


def test_exact_callbacks_see_the_canvas_after_their_frame(gpu, tmp_path):
    """exact_callbacks(true): the fragment passed to the callback of frame t holds exactly the
    frames up to t (frc.hpp:117-119) — canvas extent and bins included."""
    exe = build_driver(str(tmp_path))
    frames, _ = scroller_v1(24, dens=100)
    frames = frames.copy()
    frames[9] = 6  # a fragment break
    blob = tmp_path / "frames.bin"
    frames.tofile(blob)
    out = subprocess.run([exe, str(blob), "320", "240", "exact"], check=True, capture_output=True, text=True).stdout
    seen = {int(l.split()[1]): l.split() for l in out.strip().splitlines() if l.startswith("seen ")}
    o = O.orc_frc(frames)
    assert sorted(seen) == list(range(1, len(frames)))
    for t in range(1, len(frames)):
        f = o["frag"][t]
        idx = [i for i in range(t + 1) if o["frag"][i] == f]
        cw, ch, zx, zy = O.orc_fgm_extent(o["pos"][idx], 320, 240)
        can = np.zeros((ch, cw, 16), np.uint16)
        for i in idx:
            O.orc_fgm_blit(can, (zx, zy), frames[i], o["pos"][i])
        l = seen[t]
        assert int(l[3]) == len(idx) and [int(l[5]), int(l[6])] == [cw, ch] and [int(l[8]), int(l[9])] == [zx, zy], t
        assert int(l[11]) == poly_sum(can), t


def test_builder_collect_splice_drop_in_matches_reference(gpu, tmp_path):
    """mpb::builder::collect + splice through the drop-in headers == the reference's golden
    fgs::splice result (tests/golden/fgs_splice.npz)."""
    from remap_amd.synth import rooms_v1

    exe = build_driver(str(tmp_path))
    g = np.load(os.path.join(ROOT, "tests", "golden", "fgs_splice.npz"))
    n, w, h, rooms, dens, seed = g["gen"].tolist()
    frames, _, _ = rooms_v1(n, width=w, height=h, n_rooms=rooms, dens=dens, seed=seed)
    blob = tmp_path / "rooms.bin"
    frames.tofile(blob)
    out = subprocess.run([exe, str(blob), str(w), str(h), "splice"], check=True, capture_output=True, text=True).stdout
    lines = out.strip().splitlines()
    assert lines[0] == f"collected {int(g['ncollected'])}" and lines[1] == f"spliced {int(g['nfrag'])}"
    frag_lines = [l.split() for l in lines if l.startswith("fragment ")]
    frame_lines = [l.split() for l in lines if l.startswith("frame ")]
    assert len(frame_lines) == n
    for k, fl in enumerate(frag_lines):
        assert [int(fl[3]), int(fl[4])] == g["dims"][k].tolist()
        assert int(fl[11]) == int(g["canvas_sum"][k])
    for fl in frame_lines:
        no, k = int(fl[1]), int(fl[3])
        assert int(g["frag"][no]) == k and [int(fl[5]), int(fl[6])] == g["pos"][no].tolist()


def test_ful_files_round_trip_through_hbm(gpu, tmp_path):
    """ful::read -> device canvas -> ful::write of the drop-in layer reproduces, byte for byte,
    the files the reference's ful::write produced (tests/golden/ful_case.npz)."""
    exe = build_driver(str(tmp_path))
    g = np.load(os.path.join(ROOT, "tests", "golden", "ful_case.npz"))
    src, dst = tmp_path / "in", tmp_path / "out"
    src.mkdir()
    dst.mkdir()
    n = int(g["nfiles"])
    for k in range(n):
        (src / str(k)).write_bytes(g[f"file{k}"].tobytes())
    out = subprocess.run([exe, str(src), "1", "1", "ful", str(dst)], check=True, capture_output=True, text=True).stdout
    lines = out.strip().splitlines()
    assert lines[0] == f"read {n}"
    for k in range(n):
        w, h, zx, zy, nf = g[f"info{k}"].tolist()
        fl = lines[1 + k].split()
        assert [int(fl[2]), int(fl[3]), int(fl[5]), int(fl[6]), int(fl[8])] == [w, h, zx, zy, nf]
        assert (dst / str(k)).read_bytes() == g[f"file{k}"].tobytes()
    # a truncated file, and one whose header asks for more dots than any file holds: an error that names the
    # file — not a shortened list, not an allocation of the header's size
    whole = g["file0"].tobytes()
    for k, blob in enumerate((whole[: len(whole) // 2], np.array([1 << 40, 1 << 40], np.uint64).tobytes() + whole[16:])):
        bad = tmp_path / f"bad{k}"
        bad.mkdir()
        (bad / "0").write_bytes(whole)
        (bad / "1").write_bytes(blob)
        r = subprocess.run([exe, str(bad), "1", "1", "ful", str(dst)], capture_output=True, text=True)
        assert r.returncode != 0 and "ful::read" in r.stderr and str(bad / "1") in r.stderr, r.stderr[-500:]


def test_ful_python_round_trip(gpu, tmp_path):
    """remap_amd.ful.read uploads the golden files; write gives the same bytes back."""
    from remap_amd import ful

    g = np.load(os.path.join(ROOT, "tests", "golden", "ful_case.npz"))
    src, dst = tmp_path / "in", tmp_path / "out"
    src.mkdir()
    for k in range(int(g["nfiles"])):
        (src / str(k)).write_bytes(g[f"file{k}"].tobytes())
    frags, blobs = ful.read(str(src), with_blobs=True)
    assert frags[0].canvas.is_cuda and frags[0].step == (1, 1)
    ful.write(str(dst), frags, blobs)
    for k in range(int(g["nfiles"])):
        assert (dst / str(k)).read_bytes() == g[f"file{k}"].tobytes()


def test_collect_with_nic_compression_blobs_match_oracle(gpu, tmp_path):
    """frc::collector::collect with nic::compression (the codec main.cpp plugs in): the image
    and median blob stored with every frame equal the oracle's nic streams, frame by frame."""
    exe = build_driver(str(tmp_path))
    frames, _ = scroller_v1(30, dens=120)
    frames = frames.copy()
    frames[11] = 6  # blank frame: a 76 800-pixel run, beyond the reference codec's uint16 domain
    blob = tmp_path / "frames.bin"
    frames.tofile(blob)
    out = subprocess.run([exe, str(blob), "320", "240", "nic"], check=True, capture_output=True, text=True).stdout
    lines = out.strip().splitlines()
    blobs = {int(l.split()[1]): l.split()[2:] for l in lines if l.startswith("blob ")}
    assert sorted(blobs) == list(range(len(frames)))
    for i, f in enumerate(frames):
        med, _, _ = O.orc_kpe(f)
        assert blobs[i][0] == O.orc_nic_compress(f).hex(), i
        assert blobs[i][1] == O.orc_nic_compress(med).hex(), i
    assert lines[-1].startswith("single 1 ")


def test_fdf_filter_drop_in_matches_reference(gpu, tmp_path):
    """collect (nic::compression) -> fdf::filter through the drop-in headers: per-frame kept
    contours and masks, filtered fragments == the reference's golden fdf::filter result."""
    from test_oracle_golden import fdf_streams

    exe = build_driver(str(tmp_path))
    g = np.load(os.path.join(ROOT, "tests", "golden", "fdf_cases.npz"))
    for k, fr in enumerate(fdf_streams()):
        n, H, W = fr.shape
        blob = tmp_path / f"fdf{k}.bin"
        fr.tofile(blob)
        out = subprocess.run([exe, str(blob), str(W), str(H), "fdf"], check=True, capture_output=True, text=True).stdout
        lines = [l.split() for l in out.strip().splitlines()]
        frames = {int(l[1]): l for l in lines if l[0] == "frame"}
        frags = [l for l in lines if l[0] == "fragment"]
        want = np.unpackbits(g[f"masks{k}"])[: n * H * W].reshape(n, H, W)
        assert sorted(frames) == list(range(n))
        for i in range(n):
            l = frames[i]
            assert int(l[3]) == int(g[f"frag{k}"][i]) and int(l[5]) == int(g[f"ncontours{k}"][i]), (k, i)
            assert int(l[7]) == int(want[i].sum()), (k, i)
            h = 0
            for v in want[i].reshape(-1):
                h = (h * 1000003 + int(v)) & 0xFFFFFFFFFFFFFFFF
            assert int(l[9]) == h, (k, i)
        assert len(frags) == len(g[f"dims{k}"])
        for f, l in enumerate(frags):
            assert [int(l[3]), int(l[4])] == g[f"dims{k}"][f].tolist()
            assert int(l[11]) == int(g[f"canvas_sum{k}"][f]), (k, f)


def test_splice_drop_in_fragments_beyond_2048_keypoints(gpu, tmp_path):
    """fgs::splice through the drop-in headers with fragments whose blended canvases hold more
    than 2 048 keypoints (the snippet tables' first size): extract_single extracts again with
    larger tables.  Stream: a dense scroller run, a blank frame (fragment break), a second run
    overlapping the first; the two runs must be spliced into one fragment whose canvas equals
    the oracle's blits at the reported positions, the relative positions the oracle's offsets."""
    exe = build_driver(str(tmp_path))
    w, h = 320, 240
    frames, _ = scroller_v1(300, dens=30)
    a, b = frames[:170], frames[130:]  # 3 519 and 3 317 keypoints, 4 750 once merged
    stream = np.concatenate([a, np.full((1, h, w), 6, np.uint8), b])
    blob = tmp_path / "long.bin"
    stream.tofile(blob)
    out = subprocess.run([exe, str(blob), str(w), str(h), "splice"], check=True, capture_output=True, text=True).stdout
    lines = out.strip().splitlines()
    assert lines[0] == "collected 3" and lines[1] == "spliced 2", lines[:2]
    frag_lines = [l.split() for l in lines if l.startswith("fragment ")]
    frame_lines = [l.split() for l in lines if l.startswith("frame ")]
    assert len(frame_lines) == len(stream)
    by_frag = {}
    for fl in frame_lines:
        by_frag.setdefault(int(fl[3]), []).append((int(fl[1]), (int(fl[5]), int(fl[6]))))
    merged = [k for k, v in by_frag.items() if len(v) == len(a) + len(b)]
    assert len(merged) == 1
    k = merged[0]
    pos = dict(by_frag[k])
    # relative positions inside each run = the oracle's; between the runs = the oracle's vote
    oa, ob = O.orc_frc(a)["pos"], O.orc_frc(b)["pos"]
    for i in range(len(a)):
        assert tuple(np.array(pos[i]) - pos[0]) == tuple(oa[i] - oa[0])
    base = len(a) + 1
    for i in range(len(b)):
        assert tuple(np.array(pos[base + i]) - pos[base]) == tuple(ob[i] - ob[0])
    fl = frag_lines[k]
    cw, ch = int(fl[3]), int(fl[4])
    can = np.zeros((ch, cw, 16), np.uint16)
    for no, p in pos.items():
        O.orc_fgm_blit(can, (0, 0), stream[no], p)
    assert int(fl[11]) == poly_sum(can)


def test_pinned_link_copies_round_trip(gpu):
    """rmx_upload_pinned / rmx_download_pinned / rmx_download_2d_pinned (copies across PCIe by a few workgroups of the
    context's stream, for page-locked host memory): every size and alignment — 16-byte bodies, byte tails, unaligned
    ends — and packed rows equal to what rmx_download_2d's DMA path gives."""
    import ctypes as C

    import torch

    ctx = L.Context(320, 240)
    lib, h = ctx.lib, ctx.h
    rng = np.random.default_rng(3)
    cap = 3 << 20
    host = C.c_void_p()
    back = C.c_void_p()
    assert lib.rmx_malloc_host(h, cap, C.byref(host)) == 0 and lib.rmx_malloc_host(h, cap, C.byref(back)) == 0
    hsrc = np.ctypeslib.as_array((C.c_uint8 * cap).from_address(host.value))
    hdst = np.ctypeslib.as_array((C.c_uint8 * cap).from_address(back.value))
    hsrc[:] = rng.integers(0, 256, cap, dtype=np.uint8)
    dev = torch.zeros(cap + 64, dtype=torch.uint8, device="cuda")
    for nbytes, so, do in ((1, 0, 0), (15, 3, 5), (16, 0, 0), (4097, 16, 32), (1 << 20, 0, 16), (cap - 7, 7, 1), (cap, 0, 0)):
        dev.zero_()
        hdst[:] = 0
        assert lib.rmx_upload_pinned(h, dev.data_ptr() + do, host.value + so, nbytes) == 0
        assert lib.rmx_download_pinned(h, back.value + so, dev.data_ptr() + do, nbytes) == 0
        ctx.sync()
        assert np.array_equal(dev[do:do + nbytes].cpu().numpy(), hsrc[so:so + nbytes]), (nbytes, so, do)
        assert np.array_equal(hdst[so:so + nbytes], hsrc[so:so + nbytes]) and not hdst[so + nbytes:so + nbytes + 64].any()
    # packed rows: 300 rows of 54 bytes out of a pitch of 90
    rows, width, pitch = 300, 54, 90
    src = torch.from_numpy(rng.integers(0, 256, rows * pitch, dtype=np.uint8)).cuda()
    assert lib.rmx_download_2d_pinned(h, back.value, src.data_ptr(), pitch, width, rows) == 0
    ctx.sync()
    want = src.cpu().numpy().reshape(rows, pitch)[:, :width].reshape(-1)
    assert np.array_equal(hdst[: rows * width], want)
    assert lib.rmx_free_host(h, host) == 0 and lib.rmx_free_host(h, back) == 0
    ctx.close()
