"""Per-phase shader-clock breakdown of the best-response-only board pass (instrumented library: python -m pokerrl_amd.build --variant
timing PRL_FHP_TIMING). Usage: python scripts/phase_timing_br.py [boards] [evaluations]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import bench  # noqa: E402
import bench_br  # noqa: E402
from pokerrl_amd import _native  # noqa: E402
from pokerrl_amd.game import bet_sets  # noqa: E402
from pokerrl_amd.game import games as G  # noqa: E402

here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = _native.bind(os.path.join(here, "pokerrl_amd", "lib", "libpokerrl_hip_timing.so"))
n_boards = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 4
boards = bench.seeded_boards(n_boards, 0)
t = _native.NativeTree.for_game(G.Flop5Holdem, 20000, bet_sets.POT_ONLY, boards, _lib=L)
s = _native.NativeSolver(t, "plus", 0, engine="fused", _lib=L)
s.set_strategy(bench_br.seeded_strategy(t.n_cols - 14 * n_boards, n_boards, t.range_size, 1))
s.time_evaluations(2)
out = (ctypes.c_ulonglong * 72)()
L.prl_debug_fhp_timing.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int32]
L.prl_debug_fhp_timing(s._h, out, 1)
dev_ms, pass_ms, n_pass = s.time_evaluations(iters)
L.prl_debug_fhp_timing(s._h, out, 1)
v = np.array(list(out), np.float64)
names = ["prologue(stage+loads)", "B down/scatter", "C card scans", "D range prefix", "E up", "epilogue(store)"]
tot = v[:6].sum()
print("ms per evaluation %.3f (pass %.3f)   clocks summed over wave-0 of every board pass: %.3e" % (dev_ms / iters, pass_ms / iters, tot))
for n, x in zip(names, v[:6]):
    print("%-24s %6.2f %%   %10.0f clk per board and evaluation (both seats)" % (n, 100 * x / tot, x / n_boards / iters))
print("waiting at the barriers, clk per board-evaluation and wave (0 = board start, 1 = after B, 2 = inside C, 3 = after D, 4 = between the seats):")
for bi in range(5):
    w = v[8 + 12 * bi: 20 + 12 * bi] / n_boards / iters
    print("barrier %d: %s   mean %.0f" % (bi, " ".join("%5.0f" % x for x in w), w.mean()))
