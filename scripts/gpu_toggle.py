"""A/B of two code paths of the board pass on ONE solver (same memory, same stream): the library built with -DFHP_EXPERIMENT
(python -m pokerrl_amd.build --variant exp FHP_EXPERIMENT) switches on PrlFhpParams::exp at run time. Alternates exp = 0 / 1 and prints
the board-pass ms per iteration of each; several solvers, because the spread between allocations is larger than most effects.
Usage: python scripts/gpu_toggle.py [n_solvers] [boards]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pokerrl_amd import _native  # noqa: E402

here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = _native.bind(os.path.join(here, "pokerrl_amd", "lib", "libpokerrl_hip_exp.so"))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
boards = bench.seeded_boards(int(sys.argv[2]) if len(sys.argv) > 2 else 262144, 0)
tree = bench.fhp_tree(boards, L)
L.prl_debug_set_experiment.argtypes = [ctypes.c_void_p, ctypes.c_int32]
flags = [int(x) for x in os.environ.get("TOGGLE_FLAGS", "0,1").split(",")]
tot = {f: [] for f in flags}
for i in range(n):
    s = _native.NativeSolver(tree, "plus", 0, engine="fused", _lib=L)
    s.iterations(4)
    row = []
    for rep in range(3):
        for flag in flags:
            L.prl_debug_set_experiment(s._h, flag)
            dev_ms, pass_ms, n_pass = s.time_iterations_ex(10)
            row.append("%d:%.2f" % (flag, pass_ms / 10))
            tot[flag].append(pass_ms / 10)
    print("solver %d: %s" % (i, " ".join(row)), flush=True)
    del s
m0 = sum(tot[flags[0]]) / len(tot[flags[0]])
for f in flags:
    m = sum(tot[f]) / len(tot[f])
    print("mean exp=%d %.3f ms  (%+.2f %%)" % (f, m, 100 * (m - m0) / m0))
