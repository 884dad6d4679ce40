"""CPU-only checks: libstdc++ order emulation, the rank-selection network, the synthetic
generators, the C ABI surface (symbols only — no compute without a GPU)."""
import itertools
import os
import re

import numpy as np

import orclib as O
from remap_amd import lib as L
from remap_amd.synth import mt19937_raw, scroller_v1, upscale2

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# extra flags for the host-side checks compiled below (oracle/Makefile `asan-test`: -fsanitize=address,undefined)
HOSTCHECK_FLAGS = os.environ.get("RMX_HOSTCHECK_CXXFLAGS", "").split()


def test_umap_order_emulation_equals_libstdcxx():
    rng = np.random.default_rng(1)
    for n in [0, 1, 2, 12, 13, 14, 28, 29, 30, 58, 59, 60, 126, 127, 128, 256, 257, 258, 541, 542, 1109, 1110, 2357,
              2358, 4000]:
        for mode in range(3):
            if mode == 0:
                h = rng.integers(0, 2 ** 63, size=n, dtype=np.uint64)
            elif mode == 1:
                h = rng.integers(0, 40, size=n).astype(np.uint64)  # heavy bucket collisions
            else:
                h = (np.arange(n, dtype=np.uint64) * np.uint64(13)) + np.uint64(5)
            assert np.array_equal(O.orc_umap_order(h), O.orc_umap_order_std(h)), (n, mode)


def test_umap_bucket_growth_sequence():
    # the device table kUmapBuckets (remap_amd/csrc/kpm.hip) must be libstdc++'s sequence
    src = open(os.path.join(ROOT, "remap_amd", "csrc", "kpm.hip")).read()
    seq = [int(x) for x in re.search(r"kUmapBuckets\[\d+\] = \{([^}]*)\}", src).group(1).split(",")]
    import ctypes as C
    probe = os.path.join(ROOT, "oracle", "libremap_oracle.so")
    assert os.path.exists(probe)
    # derive the growth points from the std-based order function: the order changes structure
    # exactly when a rehash happens, so compare against the emulation for sizes around each prime
    for b in seq[:9]:
        for n in (b - 1, b, b + 1, b + 2):
            h = np.arange(n, dtype=np.uint64) * np.uint64(2654435761)
            assert np.array_equal(O.orc_umap_order(h), O.orc_umap_order_std(h)), n
    assert seq[:10] == [13, 29, 59, 127, 257, 541, 1109, 2357, 5087, 10273]


def rank4_network(a, b, c):
    """The merge used by kpe_main (remap_amd/csrc/kpe.hip merge_rows) on three sorted triples."""
    def med3(x, y, z):
        return np.maximum(np.minimum(x, y), np.minimum(np.maximum(x, y), z))
    t0 = np.maximum(np.maximum(a[0], b[0]), c[1])
    t1 = med3(a[1], b[1], c[2])
    t2 = med3(a[2], b[2], c[0])
    return med3(t0, t1, t2)


def test_rank_network_is_fourth_largest_of_nine():
    # exhaustive over 0/1 inputs (the 0-1 principle covers all totally ordered inputs) ...
    for bits in itertools.product((0, 1), repeat=9):
        v = np.array(bits).reshape(3, 3)
        s = np.sort(v, axis=1)
        got = rank4_network(s[0], s[1], s[2])
        assert got == np.sort(v.ravel())[::-1][3], bits
    # ... and random 4-bit values with ties
    rng = np.random.default_rng(0)
    v = rng.integers(0, 16, size=(20000, 3, 3))
    s = np.sort(v, axis=2)
    got = rank4_network(s[:, 0].T, s[:, 1].T, s[:, 2].T)
    want = np.sort(v.reshape(-1, 9), axis=1)[:, 5]
    assert np.array_equal(got, want)


def test_mt19937_matches_libstdcxx():
    assert mt19937_raw(42, 5).tolist() == [1608637542, 3421126067, 4083286876, 787846414, 3143890026]
    assert mt19937_raw(1234, 3).tolist() == [822569775, 2137449171, 2671936806]


def test_scroller_shapes_and_truth():
    fr, truth = scroller_v1(10)
    assert fr.shape == (10, 240, 320) and fr.dtype == np.uint8 and fr.max() < 16
    assert truth[0].tolist() == [0, 0] and (truth[:, 0] >= 0).all() and (truth[:, 0] <= 4).all()
    # frame i+1 is frame i shifted left by dx
    for i in range(1, 10):
        dx = truth[i, 0]
        assert np.array_equal(fr[i][:, : 320 - dx], fr[i - 1][:, dx:])
    assert upscale2(fr[:1]).shape == (1, 480, 640)


def test_c_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "remap_amd.h")).read()
    declared = set(re.findall(r"\b(rmx_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(L.SYMBOLS), declared ^ set(L.SYMBOLS)
    so = L.load()  # builds with hipcc if needed; loading needs no GPU
    for name in declared:
        fn = getattr(so, name)
        assert fn is not None
        # every entry point has a ctypes prototype: without one ctypes passes Python ints as
        # 32-bit C ints and a device pointer is silently truncated
        assert fn.argtypes is not None, name
        params = re.search(r"^[a-z][a-z_0-9 *]*?\b" + name + r"\s*\(([^;]*?)\)\s*;", hdr, re.S | re.M).group(1)
        n_params = 0 if params.strip() == "void" else len(params.split(","))
        if fn.argtypes or n_params == 0:
            assert len(fn.argtypes) == n_params, (name, len(fn.argtypes), n_params)
    assert so.rmx_version().startswith(b"remap_amd")
    a, b = L.luts()
    assert a.tolist() == O.orc_luts()[0].tolist() and b.tolist() == O.orc_luts()[1].tolist()


def test_extent_is_host_only_and_matches_oracle():
    import ctypes as C
    so = L.load()
    pos = np.array([[0, 0], [5, -3], [-700, 20], [900, 600]], np.int32)
    cw, ch, zx, zy = C.c_uint32(), C.c_uint32(), C.c_int32(), C.c_int32()
    assert so.rmx_fgm_extent(pos.ctypes.data, 4, 320, 240, C.byref(cw), C.byref(ch), C.byref(zx), C.byref(zy)) == 0
    assert (cw.value, ch.value, zx.value, zy.value) == O.orc_fgm_extent(pos, 320, 240)


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        return
    import pytest
    with pytest.raises(L.RemapError):
        L.Context(320, 240)


def test_ctx_create_rejects_bad_arguments_without_a_gpu():
    import ctypes as C
    so = L.load()
    h = C.c_void_p()
    bad = [(8, 240, 4, 2, 16, 64),       # too narrow
           (320, 4, 4, 2, 16, 64),       # too low
           (320, 240, 0, 2, 16, 64),     # empty grid
           (320, 240, 4, 4, 16, 64),     # more than 8 regions
           (320, 240, 4, 2, 16, 0),      # kmax 0
           (320, 240, 4, 2, 16, (1 << 22) + 1),  # kmax too large
           (64, 24, 4, 2, 16, 64)]       # 4x2 bands do not fit (kpe.hpp:84-90)
    for a in bad:
        assert so.rmx_ctx_create(0, *a, None, C.byref(h)) == -1, a
    assert so.rmx_ctx_create(0, 320, 240, 4, 2, 16, 64, None, None) == -1
    assert so.rmx_kpe_batch(None, None, 0, None, None, None) == -1
    assert so.rmx_fgm_extent(None, 0, 320, 240, None, None, None, None) == -1


def test_bitplane_3x3_pass_matches_oracle(tmp_path):
    """remap_amd/csrc/bitslice.h (the arithmetic of kpe_bs.hip: bytes <-> bit planes, rank tables on
    planes, the radix selection of the 4th largest of nine, candidate flags) compiled for the host and
    run in the kernel's arrangement (tests/cpp/bitslice_check.cpp) against the oracle's kpe
    (reference kpe.hpp:308-340): medians byte for byte, candidates = pixels whose median differs."""
    import ctypes
    import subprocess

    import orclib as O
    from remap_amd.synth import scroller_v1

    so = tmp_path / "libbitslice_check.so"
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-shared", "-fPIC", *HOSTCHECK_FLAGS, "-o", str(so),
                           os.path.join(ROOT, "tests", "cpp", "bitslice_check.cpp")])
    lib = ctypes.CDLL(str(so))
    lib.bs_planes_roundtrip.argtypes = [ctypes.c_void_p] * 3
    lib.bs_kpe_3x3.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_void_p]
    assert lib.bs_tables_ok() == 1
    rng = np.random.default_rng(1)
    for _ in range(50):
        px = rng.integers(0, 16, 32, dtype=np.uint8)
        out, pl = np.zeros(32, np.uint8), np.zeros(4, np.uint32)
        lib.bs_planes_roundtrip(px.ctypes.data, out.ctypes.data, pl.ctypes.data)
        assert (px == out).all()
        for q in range(4):
            assert int(pl[q]) == sum(int((px[i] >> q) & 1) << i for i in range(32))
    frames, _ = scroller_v1(3, dens=40)
    cases = [frames[0], frames[2], rng.integers(0, 16, (240, 320), dtype=np.uint8), rng.integers(0, 16, (48, 64), dtype=np.uint8),
             rng.integers(0, 3, (64, 32), dtype=np.uint8) * 5, np.zeros((16, 16), np.uint8)]
    for fr in cases:
        fr = np.ascontiguousarray(fr)
        Hh, Ww = fr.shape
        med, cand = np.zeros((Hh, Ww), np.uint8), np.zeros((Hh, Ww), np.uint8)
        assert lib.bs_kpe_3x3(fr.ctypes.data, Ww, Hh, med.ctypes.data, cand.ctypes.data) == 0
        omed = O.orc_kpe(fr, 1, 1, 0)[0]
        assert np.array_equal(med, omed)
        inner = np.zeros((Hh, Ww), bool)
        inner[2:Hh - 4, 2:Ww - 2] = True
        assert np.array_equal(cand.astype(bool), (omed != fr) & inner)


def test_canvas_boxes_equal_the_frame_by_frame_growth():
    """pipeline.canvas_boxes (closed form, all fragments at once) against rmx_fgm_extent's replay of
    fgm.hpp:190-233 and the oracle's, on random walks incl. ones that never leave the first step"""
    from remap_amd.pipeline import canvas_boxes

    rng = np.random.default_rng(7)
    so = L.load()
    for W, H in ((320, 240), (64, 32), (17, 5)):
        runs, pos = [], []
        for _ in range(40):
            n = int(rng.integers(1, 30))
            scale = int(rng.choice([1, 3, 40, 400]))
            p = np.cumsum(rng.integers(-scale, scale + 1, size=(n, 2)), axis=0)
            p[0] = 0
            runs.append(len(pos))
            pos += p.tolist()
        pos = np.array(pos, np.int64)
        starts = np.array(runs, np.int64)
        gw, gh, zx, zy = canvas_boxes(pos, starts, W, H)
        ends = list(starts[1:]) + [len(pos)]
        for i, (a, b) in enumerate(zip(starts, ends)):
            assert O.orc_fgm_extent(pos[a:b].astype(np.int32), W, H) == (gw[i], gh[i], zx[i], zy[i])


def test_pairwise_candidate_test_matches_oracle(tmp_path):
    """remap_amd/csrc/paircmp.h (the arithmetic of kpe_pc.hip: a pixel is a candidate unless at most three of its
    eight neighbours are greater and at least three are greater or equal — rank comparisons of adjacent pixels on
    bit planes, each shared by the two windows it belongs to) compiled for the host and run in the kernel's
    arrangement (tests/cpp/paircmp_check.cpp) against the oracle's kpe (reference kpe.hpp:308-340): candidates =
    the processed pixels whose median differs from the input, whatever the lanes at a row's ends receive."""
    import ctypes
    import subprocess

    import orclib as O
    from remap_amd.synth import scroller_v1

    so = tmp_path / "libpaircmp_check.so"
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-shared", "-fPIC", *HOSTCHECK_FLAGS, "-o", str(so),
                           os.path.join(ROOT, "tests", "cpp", "paircmp_check.cpp")])
    lib = ctypes.CDLL(str(so))
    lib.pc_kpe_cand.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_uint32]
    rng = np.random.default_rng(1)
    frames, _ = scroller_v1(3, dens=40)
    cases = [frames[0], frames[2], rng.integers(0, 16, (240, 320), dtype=np.uint8), rng.integers(0, 16, (48, 64), dtype=np.uint8),
             rng.integers(0, 3, (64, 32), dtype=np.uint8) * 5, np.zeros((16, 32), np.uint8)]
    for fr in cases:
        fr = np.ascontiguousarray(fr)
        Hh, Ww = fr.shape
        omed = O.orc_kpe(fr, 1, 1, 0)[0]
        inner = np.zeros((Hh, Ww), bool)
        inner[2:Hh - 4, 2:Ww - 2] = True
        for junk in (0, 0xffffffff, 0x5a5a1234):
            cand = np.zeros((Hh, Ww), np.uint8)
            assert lib.pc_kpe_cand(fr.ctypes.data, Ww, Hh, cand.ctypes.data, junk) == 0
            assert np.array_equal(cand.astype(bool), (omed != fr) & inner)


def test_store_data_hazard_guard_on_the_compiled_kernels(tmp_path):
    """The 16-byte median stores of kpe_bs / kpe_bs_ring / kpe_pc (buffer_store_dwordx4 with a scalar offset): no
    instruction may overwrite a store's data registers within two wait states of its issue — the corruption
    DESIGN.md 4.1e describes, which neither LLVM's hazard recogniser nor the ISA manual covers for this store
    form.  tools/check_store_hazard.py reads the compiler's output for those kernels (and is itself checked on a
    hand-made violation)."""
    import subprocess
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_store_hazard as chk

    bad = tmp_path / "bad.s"
    bad.write_text("\n_Z6kpe_pcv:\n\tbuffer_store_dwordx4 v[4:7], v1, s[36:39], s2 offen nt\n\tv_mov_b32_e32 v5, v9\n"
                   "\ts_endpgm\n\t.end_amdhsa_kernel\n")
    assert chk.check(str(bad), ["kpe_pc"]) == 1
    good = tmp_path / "good.s"
    good.write_text("\n_Z6kpe_pcv:\n\tbuffer_store_dwordx4 v[4:7], v1, s[36:39], s2 offen nt\n\ts_nop 1\n\tv_mov_b32_e32 v5, v9\n"
                    "\ts_endpgm\n\t.end_amdhsa_kernel\n")
    assert chk.check(str(good), ["kpe_pc"]) == 0
    asm = tmp_path / "device.s"
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                           os.path.join(ROOT, "remap_amd", "csrc", "api.hip"), "-o", str(asm)], stderr=subprocess.DEVNULL)
    # (fgm_blit's canvas rows and the bit-plane blit's flush store the same way)
    assert chk.check(str(asm), ["kpe_bs", "kpe_pc", "fgm_blit", "bb_flush"]) == 0
