"""
Generates tests/golden/fhp_<n>_<variant>.npz with the CPU ORACLE (oracle/prl_oracle.c): exploitability history, average-strategy
exploitability and SHA-256 of the regret / average arrays after a few iterations on a seeded Flop5Holdem board set that is too
big to re-run in every GPU test (16384 boards: 245 765 nodes, ~18 GB of oracle state, a few minutes on 8 cores).

    python tests/golden/make_fhp_golden.py [n_boards] [variant] [n_iters] [delay]

(delay > 0: the averaging weights beyond the first blends -- CFRPlus.py:65-87 -- enter the fixture; the file name then carries _d<delay>_i<n_iters>)

The tree comes from the product's host tree builder (pinned node for node to the reference's PublicTree in
tests/test_host_golden.py); everything else is the oracle. Needs no GPU.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, ".."))

import oracle  # noqa: E402
import parity_cases as pc  # noqa: E402
from helpers import env_args, h32  # noqa: E402
from pokerrl_amd import _native  # noqa: E402
from pokerrl_amd.game import bet_sets  # noqa: E402
from pokerrl_amd.game import games as G  # noqa: E402


def main(n_boards=16384, variant="plus", n_iters=3, seed=5, delay=0):
    boards = pc.fhp_boards(n_boards, seed=seed)
    args = env_args(G.Flop5Holdem, 20000, bet_sets.POT_ONLY)
    t = _native.NativeTree(G.Flop5Holdem.native_game(args), G.Flop5Holdem.native_rules(), boards)
    o = oracle.Oracle({k: t.field(k) for k in oracle.Oracle.FIELDS}, boards, 2, 52, 4, 2)
    o.cfr_reset(pc.VARIANT_ID[variant], delay)
    hist = [np.array(o.exploitability, np.float32)]
    for it in range(n_iters):
        o.cfr_iteration()
        hist.append(np.array(o.exploitability, np.float32))
        print("iteration", it + 1, hist[-1], flush=True)
    out = os.path.join(HERE, "fhp_%d_%s%s.npz" % (n_boards, variant, "_d%d_i%d" % (delay, n_iters) if delay else ""))
    np.savez(out, n_boards=n_boards, seed=seed, variant=variant, n_iters=n_iters, delay=delay, boards_sha256=h32(boards),
             expl_history=np.stack(hist), eval_avg=o.eval_avg(), regret_sha256=h32(np.asarray(o.regret)),
             avg_sha256=h32(np.asarray(o.avg)), numpy=np.__version__)
    print("wrote", out)


if __name__ == "__main__":
    a = sys.argv[1:]
    main(int(a[0]) if a else 16384, a[1] if len(a) > 1 else "plus", int(a[2]) if len(a) > 2 else 3, delay=int(a[3]) if len(a) > 3 else 0)
