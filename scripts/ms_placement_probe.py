"""Does the multi-street bench's speed depend on where its arrays land? Several solvers of one process, each timed (kept alive: different placements)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa
from bench_multistreet import runouts
from pokerrl_amd import _native
from pokerrl_amd.game import games as G
tree = _native.NativeTree.for_game(G.LimitHoldem, 48, None, runouts(4, 2, 2))
keep = []
for k in range(5):
    s = _native.NativeSolver(tree, "plus", 0)
    s.iterations(3); s.sync()
    a = s.time_iterations(10) / 10.0
    b = s.time_iterations(10) / 10.0
    print("solver %d: %.3f %.3f ms per iteration" % (k, a, b), flush=True)
    keep.append(s)
