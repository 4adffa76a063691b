"""
Generates tests/golden/fhp_br_<n>_chunked.npz with the CPU ORACLE: the exact best response of bench_br.py's seeded float32 strategy on
bench_br.py's board list at its default size (65536 boards: too big for one oracle instance), evaluated chunk by chunk exactly as
make_fhp_golden_chunked.py does for the CFR+ run (chunk instances for the boards, a trunk instance whose chance node takes the canonical
sum of all chunks' board values: orc_set_override).

    python tests/golden/make_fhp_br_golden_chunked.py [n_boards] [chunk]
    python tests/golden/make_fhp_br_golden_chunked.py --selftest          (2048 boards in chunks of 1024 = the one-piece oracle, bit for bit)

What LocalBRMaster.evaluate does after the agent query (LocalBRMaster.py:67-80): strategy -> reach -> EV + best response -> exploitability.
Needs no GPU; 65536 boards: ~20 GB of RAM, a few minutes on 8 cores.
"""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.path.insert(0, os.path.join(HERE, ".."))

import bench  # noqa: E402
import bench_br  # noqa: E402
import parity_cases as pc  # noqa: E402
from helpers import h32  # noqa: E402
from make_fhp_golden_chunked import NB, NC, group_sums, make  # noqa: E402

import oracle  # noqa: E402


def chunked_br(boards, strat, chunk):
    """-> (exploitability [2] float32, chance node's ev / ev_br [4][R] float32) of the float32 strategy `strat` ([n_cols][R], trunk columns first)"""
    n = len(boards)
    assert n % chunk == 0 and chunk % 1024 == 0
    cp = oracle.chance_prob_f32(n, 52, 2, 5)
    tt, T = make(boards[:32], chance_prob=cp)
    nt = tt.n_cols - 32 * NC
    chance = int(np.where(tt.field("kind") == 1)[0][0])
    first_board = chance + 1
    groups = []
    for c in range(n // chunk):
        t0 = time.time()
        _, O = make(boards[c * chunk:(c + 1) * chunk], chance_prob=cp)
        O.cfr_configure(1, 0)
        s = np.concatenate([strat[:nt], strat[nt + c * chunk * NC:nt + (c + 1) * chunk * NC]]).astype(np.float64)
        O.set_strategy(s, False)  # float32 values: the strategy is played as float32 (includes the reach push-down)
        O.compute_ev()
        roots = first_board + NB * np.arange(chunk)
        groups.append(group_sums(np.concatenate([O.ev[roots], O.ev_br[roots]], axis=1)))
        del O
        print("  chunk %d/%d  %.0f s" % (c + 1, n // chunk, time.time() - t0), flush=True)
    g = np.concatenate(groups)
    total = g[0].copy()
    for i in range(1, len(g)):
        total = total + g[i]
    T.cfr_configure(1, 0)
    T.set_strategy(np.concatenate([strat[:nt], strat[nt:nt + 32 * NC]]).astype(np.float64), False)
    T.set_override(chance, total[0:2], total[2:4])
    T.compute_ev()
    return np.array(T.exploitability, np.float32), total


def selftest():
    boards = pc.fhp_boards(2048, seed=5)
    t, o = make(boards)
    strat = bench_br.seeded_strategy(t.n_cols - 2048 * NC, 2048, t.range_size, 7)
    expl, total = chunked_br(boards, strat, 1024)
    o.cfr_configure(1, 0)
    o.set_strategy(strat.astype(np.float64), False)
    o.compute_ev()
    chance = int(np.where(t.field("kind") == 1)[0][0])
    assert np.array_equal(expl, np.array(o.exploitability, np.float32)), (expl, o.exploitability)
    assert np.array_equal(total[0:2], o.ev[chance]) and np.array_equal(total[2:4], o.ev_br[chance])
    print("selftest ok: the chunked evaluation equals the one-piece oracle (2048 boards)", expl)


def main(n_boards, chunk):
    boards = bench.seeded_boards(n_boards, 0)  # bench_br.py's rank-0 board list
    t, _ = make(boards[:32])
    nt, R = t.n_cols - 32 * NC, t.range_size
    strat = bench_br.seeded_strategy(nt, n_boards, R, 1)  # bench_br.py's rank-0 strategy
    expl, total = chunked_br(boards, strat, chunk)
    out = os.path.join(HERE, "fhp_br_%d_chunked.npz" % n_boards)
    np.savez(out, n_boards=n_boards, board_seed=0, strategy_seed=1, chunk=chunk, boards_sha256=h32(boards), strategy_sha256=h32(strat),
             exploitability=expl, chance_ev=total[0:2], chance_ev_br=total[2:4], numpy=np.__version__)
    print("wrote", out, expl)


if __name__ == "__main__":
    a = sys.argv[1:]
    if a and a[0] == "--selftest":
        selftest()
    else:
        main(int(a[0]) if a else 65536, int(a[1]) if len(a) > 1 else 8192)
