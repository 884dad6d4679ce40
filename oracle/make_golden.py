#!/usr/bin/env python3
"""make_golden.py — generate tests/golden/*.npz from the REFERENCE itself.

Runs only where oracle/_ref/*.so exist (built by oracle/build_ref.sh from /root/reference).
Every expected value stored here was produced by the reference's own code:
  tier 1 (libref_strict.so, unmodified headers): LUTs, hashes, kpm grid match + tickets,
          kpm cellular match, fgm fragment blit/extent/blend/merge;
  tier 2 (libref_full.so, patched temporary copy — see build_ref.sh): kpe tables/medians,
          frc::collector::collect streams, fgs::splice.
Fixtures hold inputs (or the seed of the committed generator remap_amd/synth.py plus a
checksum of the generated input) and expected outputs only — never reference source.

    python oracle/make_golden.py          # rewrites tests/golden/
    python oracle/make_golden.py ful      # only the fragment-file fixture (ful_case.npz)
    python oracle/make_golden.py nic      # only the codec fixture (nic_cases.npz)
    python oracle/make_golden.py fdf      # only the foreground-filter fixture (fdf_cases.npz)
"""
from __future__ import annotations

import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import orclib as O  # noqa: E402
from remap_amd.synth import rooms_v1, scroller_v1  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def structured_image(rng, H, W, block=6):
    base = rng.integers(0, 16, size=(H // block + 1, W // block + 1), dtype=np.uint8)
    img = np.kron(base, np.ones((block, block), np.uint8))[:H, :W].copy()
    # sprinkle single pixels and thin lines so weight-1 and weight-2 keypoints both occur
    for _ in range(H * W // 150):
        img[rng.integers(0, H), rng.integers(0, W)] = rng.integers(0, 16)
    return img


def golden_ful():
    """Fragment files written by the reference's ful::write (ful.hpp:10-46, tier 1): two small
    fragments, one grown to the left/top with blobs on its frames, one fresh."""
    import tempfile

    assert O.ref_strict() is not None, "run oracle/build_ref.sh first"
    rng = np.random.default_rng(77)
    W, H = 16, 8
    a = O.RefFragment(W, H)
    pos = [(0, 0), (3, 1), (-5, -2), (17, 4)]
    for i, p in enumerate(pos):
        img = rng.integers(0, 16, size=(H, W), dtype=np.uint8)
        a.blit_packed(p, img, 100 + i, rng.integers(0, 256, size=int(rng.integers(0, 9)), dtype=np.uint8).tobytes(),
                      rng.integers(0, 256, size=int(rng.integers(1, 6)), dtype=np.uint8).tobytes())
    b = O.RefFragment(W, H)
    b.blit((0, 0), rng.integers(0, 16, size=(H, W), dtype=np.uint8), 7)
    out = {}
    with tempfile.TemporaryDirectory() as d:
        O.ref_ful_write(d, [a, b])
        names = sorted(os.listdir(d), key=int)
        assert names == ["0", "1"]
        for k, f in enumerate((a, b)):
            out[f"file{k}"] = np.frombuffer(open(os.path.join(d, str(k)), "rb").read(), np.uint8)
            out[f"dots{k}"] = f.dots()
            out[f"info{k}"] = np.array(f.info(), np.int64)  # w, h, zero_x, zero_y, frames
            nums, fpos = f.frames()
            out[f"numbers{k}"] = nums
            out[f"pos{k}"] = fpos
            bl = f.blobs()
            out[f"blob_sizes{k}"] = np.array([[len(x), len(y)] for x, y in bl], np.int64)
            out[f"blob_bytes{k}"] = np.frombuffer(b"".join(x + y for x, y in bl) or b"\0", np.uint8)
        # and the reference reads its own files back to the same fragments (step (1, 1))
        back = O.ref_ful_read(d)
        assert len(back) == 2 and np.array_equal(back[0].dots(), out["dots0"]) and back[0].info() == a.info()
    out["nfiles"] = np.array(2)
    np.savez_compressed(os.path.join(OUT, "ful_case.npz"), **out)
    print("ful fixture:", {k: v.shape for k, v in out.items() if k.startswith("file")})


def nic_inputs():
    """Inputs of the nic fixture (all inside the reference's domain: runs <= 65535 pixels,
    literals <= 16383 — outside it nic::compress corrupts its heap)."""
    rng = np.random.default_rng(4242)
    fr, _ = scroller_v1(4, dens=200)
    frd, _ = scroller_v1(2, dens=40)
    a = np.zeros((20, 100), np.uint8)
    a[:, ::4] = 5
    cases = [fr[0], fr[3], frd[1], rng.integers(0, 16, size=(64, 200), dtype=np.uint8),
             rng.integers(0, 2, size=(48, 300), dtype=np.uint8), a, np.full((100, 600), 7, np.uint8),
             np.kron(rng.integers(0, 16, size=(10, 40), dtype=np.uint8), np.ones((1, 7), np.uint8)),
             np.kron(rng.integers(0, 16, size=(6, 5), dtype=np.uint8), np.ones((1, 300), np.uint8)),
             np.full((1, 65535), 3, np.uint8), np.array([[1]], np.uint8), np.array([[1, 1]], np.uint8),
             np.array([[1, 1, 1]], np.uint8), np.array([[1, 2, 2, 2]], np.uint8), np.array([[1, 2, 3, 3, 3, 4]], np.uint8)]
    lit = rng.integers(0, 16, size=16383, dtype=np.uint8)  # the longest literal the header can express
    lit[1:][lit[1:] == lit[:-1]] ^= 1
    cases.append(np.concatenate([lit, [lit[-1] ^ 2] * 3]).astype(np.uint8)[None])
    for _ in range(200):
        n = int(rng.integers(1, 400))
        p = rng.random() ** 2
        x = np.zeros(n, np.uint8)
        v = int(rng.integers(0, 16))
        for i in range(n):
            if rng.random() < p:
                v = int(rng.integers(0, 16))
            x[i] = v
        cases.append(x[None])
    return cases


def golden_nic():
    """nic::compress of the reference (nic.hpp:8-101, tier 1, unmodified) on nic_inputs()."""
    assert O.ref_strict() is not None, "run oracle/build_ref.sh first"
    cases = nic_inputs()
    out = {"ncases": np.array(len(cases)), "shapes": np.array([c.shape for c in cases], np.int64),
           "pixels": np.concatenate([c.reshape(-1) for c in cases])}
    packs = [O.ref_nic_compress(c) for c in cases]
    for c, pk in zip(cases, packs):
        assert np.array_equal(O.ref_nic_decompress(pk, c.shape[1], c.shape[0]), c)
    out["sizes"] = np.array([len(pk) for pk in packs], np.int64)
    out["packed"] = np.frombuffer(b"".join(packs), np.uint8)
    np.savez_compressed(os.path.join(OUT, "nic_cases.npz"), **out)
    print("nic fixture:", len(cases), "cases,", int(out["sizes"].sum()), "compressed bytes")


def fdf_inputs():
    """Streams of the fdf fixture (generator parameters only; remap_amd/synth.py is committed).
    Widths are multiples of 32: fde.hpp's aligned 256-bit loads need it under g++."""
    from remap_amd.synth import with_sprites

    a, _ = scroller_v1(24, width=128, height=64, dens=30)
    a = with_sprites(a, n_sprites=3, size=11, seed=5)
    a[10] = 6  # blank frame: fragment break
    b, _, _ = rooms_v1(60, width=128, height=96, n_rooms=2, dens=40, seed=3)
    b = with_sprites(b, n_sprites=2, size=9, seed=9)
    c, _ = scroller_v1(8, width=320, height=240, dens=200)
    c = with_sprites(c, n_sprites=3, size=15, seed=1)
    return [a, b, c]


def golden_fdf():
    """frc collect + fdf::filter of the reference (fdf.hpp:40-89, tier 2 / libref_fdf.so)."""
    assert O.ref_fdf_lib() is not None, "run oracle/build_ref.sh first (needs AVX-512VL/BW)"
    out = {}
    for k, fr in enumerate(fdf_inputs()):
        r = O.ref_fdf(fr)
        out[f"input_sha256_{k}"] = np.frombuffer(bytes.fromhex(sha(fr)), np.uint8)
        out[f"shape{k}"] = np.array(fr.shape)
        out[f"masks{k}"] = np.packbits(r["masks"] != 0)
        out[f"ncontours{k}"] = r["ncontours"]
        out[f"frag{k}"] = r["frag"]
        out[f"dims{k}"] = r["dims"]
        out[f"canvas_sum{k}"] = r["canvas_sum"]
        print("fdf case", k, fr.shape, "fragments", r["nfrag"], "mask pixels", int((r["masks"] != 0).sum()),
              "contours/frame", float(r["ncontours"].mean()))
    out["ncases"] = np.array(len(fdf_inputs()))
    np.savez_compressed(os.path.join(OUT, "fdf_cases.npz"), **out)


def main():
    if sys.argv[1:] == ["fdf"]:
        return golden_fdf()
    if sys.argv[1:] == ["ful"]:
        return golden_ful()
    if sys.argv[1:] == ["nic"]:
        return golden_nic()
    assert O.ref_strict() is not None and O.ref_full() is not None, "run oracle/build_ref.sh first"
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(20261003)
    S = O.ref_strict()

    # ---- LUTs and hashes (tier 1) -------------------------------------------------------------
    n2o = np.zeros(16, np.uint8)
    o2n = np.zeros(16, np.uint8)
    S.ref_luts(O.ptr(n2o), O.ptr(o2n))
    codes = rng.integers(0, 256, size=(8, 13), dtype=np.uint8)
    offs = np.array([[0, 0], [1, 0], [-1, 0], [3, -7], [-320, 240], [32767, -32767], [5, 5], [-2, 117]], np.int32)
    code_h = [int(S.ref_hash_code(O.ptr(np.ascontiguousarray(c)))) for c in codes]
    off_h = [int(S.ref_hash_offset(int(x), int(y))) for x, y in offs]
    json.dump({"native_to_ordered": n2o.tolist(), "ordered_to_native": o2n.tolist(),
               "hash_codes": codes.tolist(), "hash_code_values": code_h,
               "hash_offsets": offs.tolist(), "hash_offset_values": off_h},
              open(os.path.join(OUT, "luts_hashes.json"), "w"), indent=1)

    # ---- kpe (tier 2) ---------------------------------------------------------------------------
    kpe = {}
    imgs = [structured_image(rng, 48, 128), structured_image(rng, 64, 160, 9), structured_image(rng, 37, 132, 5),
            rng.integers(0, 16, size=(24, 64), dtype=np.uint8)]
    fr, _ = scroller_v1(3, dens=40)
    imgs.append(fr[2])
    # hand-made 5x5 facts: the survey's verified patch at several places
    hand = np.full((40, 128), 6, np.uint8)
    patch = np.array([[6, 6, 6, 6, 6], [6, 2, 2, 2, 6], [6, 2, 5, 2, 6], [6, 2, 2, 2, 6], [6, 6, 6, 6, 6]], np.uint8)
    for (y, x) in [(10, 10), (20, 80), (4, 60), (30, 100), (33, 3)]:
        hand[y - 2:y + 3, x - 2:x + 3] = patch
    imgs.append(hand)
    for i, img in enumerate(imgs):
        for single in (0, 1):
            if single and img.shape[1] < 64:
                continue
            # the 4x2 band arithmetic (kpe.hpp:84-90) leaves the frame below 112x30: the
            # reference then reads outside its scratch matrix, so such sizes are not vectors
            if not single and (img.shape[1] < 112 or img.shape[0] < 30):
                continue
            med, desc, k = O.ref_kpe(img, bool(single))
            kpe[f"img{i}"] = img
            kpe[f"med{i}_{single}"] = med
            kpe[f"desc{i}_{single}"] = desc.view(np.uint8).reshape(-1, 18)
    np.savez_compressed(os.path.join(OUT, "kpe_cases.npz"), **kpe)

    # ---- kpm grid variant (tier 1) ---------------------------------------------------------------
    kpm = {}
    ci = 0
    for dens, nfr in ((40, 10), (200, 6), (25, 5)):
        fr, _ = scroller_v1(nfr, dens=dens)
        tabs = [O.ref_kpe(f)[1] for f in fr]
        for i in range(1, nfr):
            m, d, t = O.ref_kpm_grid(tabs[i - 1], tabs[i])
            kpm[f"prev{ci}"] = tabs[i - 1].view(np.uint8).reshape(-1, 18)
            kpm[f"cur{ci}"] = tabs[i].view(np.uint8).reshape(-1, 18)
            kpm[f"res{ci}"] = np.array([m, d[0], d[1]], np.int32)
            kpm[f"tick{ci}"] = t.view(np.uint32).reshape(8, 3, 4)
            ci += 1
    # degenerate pairs: empty tables, one active region only, unrelated frames
    empty = np.zeros(0, O.DESC_DTYPE)
    some = O.ref_kpe(scroller_v1(2, dens=40)[0][1])[1]
    one_region = some[some["region_mask"] == 1]
    other = O.ref_kpe(structured_image(rng, 240, 320, 11))[1]
    for a, b in ((empty, some), (some, empty), (empty, empty), (one_region, one_region), (some, other), (some, some)):
        m, d, t = O.ref_kpm_grid(a, b)
        kpm[f"prev{ci}"] = a.view(np.uint8).reshape(-1, 18)
        kpm[f"cur{ci}"] = b.view(np.uint8).reshape(-1, 18)
        kpm[f"res{ci}"] = np.array([m, d[0], d[1]], np.int32)
        kpm[f"tick{ci}"] = t.view(np.uint32).reshape(8, 3, 4)
        ci += 1
    kpm["n"] = np.array(ci)
    np.savez_compressed(os.path.join(OUT, "kpm_grid_cases.npz"), **kpm)

    # ---- fgm fragment (tier 1) -------------------------------------------------------------------
    fgm = {}
    fr, _ = scroller_v1(8, width=64, height=32, dens=30)
    for case, poss in enumerate([
        [(0, 0), (3, 0), (5, 1), (70, 2), (-3, -1), (-65, 33), (10, -40), (64, 32)],
        [(0, 0), (0, 0), (1, 1), (2, 2), (-1, -1), (-64, -32), (128, 64), (-1, 63)],
    ]):
        f = O.RefFragment(64, 32)
        trace = []
        for i, p in enumerate(poss):
            f.blit(p, fr[i], i)
            trace.append(f.info()[:4])
        img, mask = f.blend()
        fgm[f"pos{case}"] = np.array(poss, np.int32)
        fgm[f"trace{case}"] = np.array(trace, np.int64)
        fgm[f"dots{case}"] = f.dots()
        fgm[f"blend_img{case}"] = img
        fgm[f"blend_mask{case}"] = mask
    # merge: dst.blit(pos, std::move(src)) then normalize (fgs.hpp:170-171)
    a = O.RefFragment(64, 32)
    b = O.RefFragment(64, 32)
    for i in range(4):
        a.blit((3 * i, i), fr[i], i)
        b.blit((-2 * i, 2 * i), fr[4 + i], 4 + i)
    fgm["merge_a_dots"] = a.dots()
    fgm["merge_b_dots"] = b.dots()
    fgm["merge_a_info"] = np.array(a.info()[:4], np.int64)
    fgm["merge_b_info"] = np.array(b.info()[:4], np.int64)
    az = a.info()[2:4]
    a.merge((az[0] + 37, az[1] - 11), b)
    a.normalize()
    fgm["merge_at"] = np.array([37, -11], np.int32)
    fgm["merge_dots"] = a.dots()
    fgm["merge_info"] = np.array(a.info()[:4], np.int64)
    nums, fpos = a.frames()
    fgm["merge_frame_numbers"] = nums
    fgm["merge_frame_pos"] = fpos
    fgm["frames"] = fr
    np.savez_compressed(os.path.join(OUT, "fgm_cases.npz"), **fgm)

    # ---- frc streams (tier 2): config 1 and a dense / broken stream -------------------------------
    frc = {}
    for name, (n, dens) in {"config1": (100, 200), "dense": (60, 40)}.items():
        fr, truth = scroller_v1(n, dens=dens)
        if name == "dense":
            fr = fr.copy()
            fr[20] = 6   # blank frame -> fragment break
            fr[41] = 6
        r = O.ref_frc(fr)
        frc[f"{name}_n"] = np.array([n, dens])
        frc[f"{name}_input_sha256"] = np.frombuffer(bytes.fromhex(sha(fr)), np.uint8)
        frc[f"{name}_frag"] = r["frag"]
        frc[f"{name}_pos_norm"] = r["pos_norm"]
        frc[f"{name}_dims"] = r["dims"]
        frc[f"{name}_canvas_sum"] = r["canvas_sum"]
        frc[f"{name}_kentries"] = r["kentries"]
        frc[f"{name}_truth"] = truth
    np.savez_compressed(os.path.join(OUT, "frc_streams.npz"), **frc)

    # ---- kpm cellular variant (tier 1) + fgs::splice (tier 2) -------------------------------------
    cell = {}
    ci = 0
    fr, room, cam = rooms_v1(90, width=128, height=96, n_rooms=2, dens=40, seed=3)
    # fragments as the splice stage sees them: blend of a canvas, single-region tables
    segs = []
    start = 0
    for i in range(len(fr) + 1):
        if i == len(fr) or room[i] < 0:
            if i - start >= 2:
                segs.append((start, i))
            start = i + 1
    frags = []
    for (s, e) in segs[:4]:
        f = O.RefFragment(128, 96)
        for i in range(s, e):
            f.blit((int(cam[i, 0] - cam[s, 0]), int(cam[i, 1] - cam[s, 1])), fr[i], i)
        f.normalize()
        img, mask = f.blend()
        _, desc, _ = O.ref_kpe(img, True)
        frags.append((img, mask, desc))
    for i in range(len(frags)):
        for j in range(len(frags)):
            if i == j:
                continue
            m, v = O.ref_kpm_cell(frags[i][2], frags[i][1], frags[j][2], frags[j][1].shape)
            cell[f"res{ci}"] = np.array([m, v["dx"], v["dy"], v["count"], i, j], np.int64)
            ci += 1
    for i, (img, mask, desc) in enumerate(frags):
        cell[f"img{i}"] = img
        cell[f"mask{i}"] = mask
        cell[f"desc{i}"] = desc.view(np.uint8).reshape(-1, 18)
    cell["nfrag"] = np.array(len(frags))
    cell["npairs"] = np.array(ci)
    np.savez_compressed(os.path.join(OUT, "kpm_cell_cases.npz"), **cell)

    sp = O.ref_frc_splice(fr)
    np.savez_compressed(os.path.join(OUT, "fgs_splice.npz"), input_sha256=np.frombuffer(bytes.fromhex(sha(fr)), np.uint8),
                        gen=np.array([90, 128, 96, 2, 40, 3]), nfrag=np.array(sp["nfrag"]),
                        ncollected=np.array(sp["ncollected"]), frag=sp["frag"], pos=sp["pos"], dims=sp["dims"],
                        canvas_sum=sp["canvas_sum"])
    golden_ful()
    golden_nic()
    if O.ref_fdf_lib() is not None:
        golden_fdf()
    tot = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print("golden fixtures written to", OUT, f"({tot / 1024:.0f} KiB)")
    print("fgs: collected", sp["ncollected"], "-> spliced", sp["nfrag"], "dims", sp["dims"].tolist())
    print("cell pairs:", [cell[f"res{i}"].tolist() for i in range(ci)])


def golden_rooms():
    """BASELINE.json configs[4] in small: a 2 000-frame rooms-v1 stream at 320x240 through the REFERENCE's
    collect (frc.hpp) and splice (fgs.hpp:187-213): 80 fragments, 16 merges — the greedy loop's
    select_match order and the re-matching after every merge are exercised sixteen times — plus the
    reference's cellular kpm::match votes (kpm.hpp:371-393, unmodified tier) for all 3 160 initial pairs."""
    gen = (2000, 320, 240, 20, 200, 11)
    n, w, h, rooms, dens, seed = gen
    fr, _, _ = rooms_v1(n, width=w, height=h, n_rooms=rooms, dens=dens, seed=seed)
    sp = O.ref_frc_splice(fr)
    col = O.ref_frc(fr)
    nf = int(col["frag"].max()) + 1
    assert nf == sp["ncollected"]
    tabs = []
    for f in range(nf):
        idx = np.nonzero(col["frag"] == f)[0]
        rf = O.RefFragment(w, h)
        for i in idx:
            rf.blit(col["pos_norm"][i], fr[i], int(i))
        assert rf.info()[:2] == tuple(col["dims"][f].tolist())
        img, mask = rf.blend()
        _, desc, _ = O.ref_kpe(img, True)
        tabs.append((mask, desc))
    pairs = [(i, j) for i in range(nf) for j in range(i + 1, nf)]
    votes = np.zeros((len(pairs), 4), np.int32)
    for k, (i, j) in enumerate(pairs):
        m, v = O.ref_kpm_cell(tabs[i][1], tabs[i][0], tabs[j][1], tabs[j][0].shape)
        votes[k] = (v["dx"], v["dy"], v["count"], 1) if m else (0, 0, 0, 0)
    np.savez_compressed(os.path.join(OUT, "fgs_splice_rooms.npz"), input_sha256=np.frombuffer(bytes.fromhex(sha(fr)), np.uint8),
                        gen=np.array(gen), nfrag=np.array(sp["nfrag"]), ncollected=np.array(sp["ncollected"]), frag=sp["frag"],
                        pos=sp["pos"], dims=sp["dims"], canvas_sum=sp["canvas_sum"], collected_frag=col["frag"],
                        collected_dims=col["dims"], votes=votes, kcounts=np.array([len(t[1]) for t in tabs]))
    print("rooms: collected", sp["ncollected"], "-> spliced", sp["nfrag"], "valid initial votes", int(votes[:, 3].sum()), "of", len(pairs))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "rooms":
        golden_rooms()
    else:
        main()
        golden_rooms()
