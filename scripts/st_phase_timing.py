"""Per-phase shader clocks of the last street's pass (per-street engine), from a PRL_ST_TIMING build:
    python -m pokerrl_amd.build --variant sttiming PRL_ST_TIMING ; python scripts/st_phase_timing.py [flops turns rivers]"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import torch  # noqa: E402,F401
from bench_multistreet import runouts  # noqa: E402
from pokerrl_amd import _native  # noqa: E402
from pokerrl_amd.game import games as G  # noqa: E402

L = _native.bind(os.path.join(ROOT, "pokerrl_amd", "lib", "libpokerrl_hip_sttiming.so"))
f, t, r = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (4, 2, 2)
tree = _native.NativeTree.for_game(G.LimitHoldem, 48, None, runouts(f, t, r), _lib=L)
s = _native.NativeSolver(tree, "plus", 0, engine="auto", _lib=L)
s.iterations(3)
out = np.zeros(8, np.uint64)
L.prl_debug_st_timing.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int32]
assert L.prl_debug_st_timing(s._h, out.ctypes.data_as(ctypes.c_void_p), 1) == 0
n = 10
ms = s.time_iterations(n)
assert L.prl_debug_st_timing(s._h, out.ctypes.data_as(ctypes.c_void_p), 0) == 0
names = ["0 set-up (table entry, plan, root reach)", "1 opponent columns + reach walk", "2 barrier after the walk", "3 scans C + D", "4 value walk + updates",
         "5 row stores / turn-around", "6 barrier at the instance start", "7 -"]
tot = float(out.sum())
print("%d iterations, %.3f ms each; shader clocks of wave 0 per phase (summed over workgroups and passes):" % (n, ms / n))
for k in range(7):
    print("  %-45s %6.2f %%" % (names[k], 100.0 * float(out[k]) / tot))
