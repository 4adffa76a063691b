"""Diagnostics: is the board pass's fast / slow mode (27.4 vs 30.5+ ms per iteration at 262144 boards) a property of the process, of the
allocation, or of the moment? One process builds and times the same solve several times; every solver is timed three times (reset in
between: same memory, same stream). Usage: python scripts/gpu_mode_probe.py [n]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from pokerrl_amd import _native  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4
_L = _native.bind(os.environ["PROBE_LIB"]) if os.environ.get("PROBE_LIB") else None  # another build of the library
boards = bench.seeded_boards(int(os.environ.get("PROBE_BOARDS", "262144")), 0)
tree = bench.fhp_tree(boards, _L)
keep = []
for i in range(n):
    import time
    t0 = time.perf_counter()
    s = _native.NativeSolver(tree, "plus", 0, engine="fused", _lib=_L)
    s.sync()
    t_create = time.perf_counter() - t0
    out = []
    for rep in range(int(os.environ.get("PROBE_REPS", "3"))):
        s.reset()
        s.iterations(4)
        dev_ms, pass_ms, n_pass = s.time_iterations_ex(10)
        out.append(pass_ms / 10)
    print("solver %d: board pass %s ms per iteration, whole iteration %.3f (solver created in %.1f s)" % (i, " ".join("%.3f" % x for x in out), dev_ms / 10, t_create), flush=True)
    if os.environ.get("PROBE_KEEP"):
        keep.append(s)  # the next solver lands in other memory
    else:
        del s
