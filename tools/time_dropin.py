"""Throughput of the C++ drop-in frc::collector::collect (host feed -> HBM -> callbacks).
usage: python tools/time_dropin.py [frames] [modes...]   REMAP_BLOCK_FRAMES=n: frames per block; REMAP_PHASES=1: where collect spends its time"""
import os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from remap_amd import lib as L
from remap_amd.synth import scroller_v1
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
modes = sys.argv[2:] or ["time", "timenic", "timeview", "timeviewnic"]
L.load()
tmp = tempfile.mkdtemp()
exe = os.path.join(tmp, "drop_in")
subprocess.check_call(["g++", "-std=c++20", "-O2", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "include", "remap", "compat"),
                       os.path.join(ROOT, "tests", "cpp", "drop_in.cpp"), "-L" + os.path.join(ROOT, "remap_amd"), "-lremap_amd",
                       "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + os.path.join(ROOT, "remap_amd"), "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
fr, _ = scroller_v1(n, dens=200)
blob = os.path.join(tmp, "f.bin")
fr.tofile(blob)
for mode in modes:
    r = subprocess.run([exe, blob, "320", "240", mode], check=True, capture_output=True, text=True, env=dict(os.environ, REMAP_PROFILE="1"))
    print(r.stdout.strip(), flush=True)
    print(r.stderr.strip() if os.environ.get("REMAP_PHASES") else "", end="", flush=True)
