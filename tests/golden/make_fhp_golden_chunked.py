"""
Generates tests/golden/fhp_<n>_<variant>_chunked.npz with the CPU ORACLE for board sets too big for one oracle instance (262144 boards =
bench.py's default size: ~290 GB of oracle state in one piece): the boards are evaluated CHUNK BY CHUNK, the trunk in a second small
instance whose chance node takes the canonical sum of all chunks' board values from outside (oracle: orc_set_override).

    python tests/golden/make_fhp_golden_chunked.py [n_boards] [n_iters] [chunk] [workdir] [plus|linear|vanilla]      (delay 0)
    python tests/golden/make_fhp_golden_chunked.py --whole-game [n_iters] [chunk] [workdir] [variant]    the WHOLE Flop5Holdem game through its 134 459 suit
        classes (prl_solver_create_weighted: multiplicities in the chance weights, orbit-mean chance values) -> tests/golden/fhp_whole_game_<variant>_chunked.npz

Per half-iteration of _CFRBase.iteration (_CFRBase.py:122-134) -- EVs, regrets + strategy of seat p, reach, average of seat p --:
  every chunk instance: trunk strategy from the trunk instance, its own boards' state from disk; update_reach; compute_ev (the board values
      depend on the reach at the chance node only); the 1024-board group sums of the board roots' ev / ev_br (canonical order: blocks of
      32, groups of 32 blocks); regrets / strategy / average of seat p's BOARD nodes; state back to disk
  trunk instance: chance node := running sum of all groups in global order; compute_ev; regrets / strategy / reach / average of seat p's
      trunk nodes.
Vanilla / Linear CFR: the average of seat p needs the reach AFTER seat p's update, boards included (VanillaCFR.py:40-55, LinearCFR.py:41-57), and the
boards' reach needs the trunk's new strategy, which needs every chunk's values first: a chunk adds seat p's strategy to its averages at the START of its
next visit (the reach it computes there is the one the reference used: nothing changed in between), the last visit being the closing evaluation.
The evaluation that closes iteration t is the one iteration t + 1 starts with. The chunked run is checked against the one-piece oracle at
a size both can do (`--selftest`: 2048 boards in chunks of 1024, every array bit for bit).
Needs no GPU, ~60 GB of scratch disk and about an hour on 8 cores for 262144 boards x 2 iterations.
"""
import hashlib
import os
import sys
import time

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

NB = 15  # nodes per board subtree (FHP15)
NC = 14  # action columns per board subtree


def make(boards, chance_prob=None):
    args = env_args(G.Flop5Holdem, 20000, bet_sets.POT_ONLY)
    t = _native.NativeTree(G.Flop5Holdem.native_game(args), G.Flop5Holdem.native_rules(), boards)
    o = oracle.Oracle({k: t.field(k) for k in oracle.Oracle.FIELDS}, boards, 2, 52, 4, 2, chance_prob=chance_prob)
    return t, o


def group_sums(vals):
    """[n_boards][4][R] float32 (ev seat 0 / 1, ev_br seat 0 / 1 of the board roots) -> [ceil(n_boards / 1024)][4][R]: running adds over blocks of 32
    boards, then over the 32 blocks of a group (the canonical chance sum of DESIGN.md, levels 0 and 1). A ragged tail (the last chunk of a board list
    that is no multiple of 1024) makes a last group of fewer blocks and a last block of fewer boards: the same running adds over what there is."""
    n = vals.shape[0]
    if n % 1024:
        full = n - n % 1024
        out = [group_sums(vals[:full])] if full else []
        tail = vals[full:]
        gsum = None
        for b0 in range(0, len(tail), 32):
            bsum = tail[b0].copy()
            for i in range(b0 + 1, min(b0 + 32, len(tail))):
                bsum = bsum + tail[i]
            gsum = bsum if gsum is None else gsum + bsum
        return np.concatenate(out + [gsum[None]])
    v = vals.reshape(n // 32, 32, *vals.shape[1:])
    blk = v[:, 0].copy()
    for i in range(1, 32):
        blk = blk + v[:, i]
    g = blk.reshape(n // 1024, 32, *vals.shape[1:])
    out = g[:, 0].copy()
    for i in range(1, 32):
        out = out + g[:, i]
    return out


class Chunked:
    def __init__(self, boards, chunk, workdir, variant="plus", mult=None, sym_class=None):
        """mult / sym_class: weighted boards (suit classes with multiplicities) and the hands' suit classes for the orbit-mean chance values"""
        self.boards, self.chunk, self.workdir = boards, chunk, workdir
        self.mult, self.sym_class = (None if mult is None else np.asarray(mult, np.int64)), sym_class
        self.variant = variant
        self.vcode = {"vanilla": 0, "plus": 1, "linear": 2}[variant]
        self.pending = None  # Vanilla / Linear: (iteration, seat) whose strategy the chunks still have to add to their averages
        self.n = len(boards)
        assert chunk % 1024 == 0 and (self.n % chunk == 0 or mult is not None)
        self.n_chunks = -(-self.n // chunk)
        self.total = self.n if mult is None else int(self.mult.sum())  # the boards the listed ones stand for
        self.cp = oracle.chance_prob_f32(self.total, 52, 2, 5)
        self.tt, self.T = make(boards[:32], chance_prob=self.cp)       # trunk instance (its own boards are never looked at)
        self.ct, self.O = make(boards[:chunk], chance_prob=self.cp)    # chunk instance, re-used for every chunk (same tree shape; boards swapped below)
        self.nt = self.ct.n_cols - chunk * NC                          # trunk columns come first
        kind = self.ct.field("kind")
        self.chance = int(np.where(kind == 1)[0][0])
        self.first_board = self.chance + 1
        assert self.ct.n_nodes == self.first_board + chunk * NB and int(np.where(self.tt.field("kind") == 1)[0][0]) == self.chance
        self.updated = [False, False]  # seats whose strategy comes from regret matching (else the uniform fill)
        if sym_class is not None:
            self.T.set_symmetrize(sym_class)  # applies to the chance node's override: the weighted sum of all chunks (oracle: compute_ev)
        self.T.cfr_reset(self.vcode, 0)
        self.hist = []

    def _path(self, c, name):
        return os.path.join(self.workdir, "chunk%03d_%s.npy" % (c, name))

    def _chunk_instance(self, c):
        """a fresh oracle on chunk c's boards (the showdown plans belong to the boards) with its state from disk"""
        del self.O
        _, self.O = make(self.boards[c * self.chunk:(c + 1) * self.chunk], chance_prob=self.cp)
        O = self.O
        if self.mult is not None:
            O.set_board_weights(self.mult[c * self.chunk:(c + 1) * self.chunk], total=self.total)
        O.cfr_configure(self.vcode, 0)  # uniform strategy, zero regrets / averages (no evaluation yet)
        if os.path.exists(self._path(c, "regret")):
            O.regret[self.nt:] = np.load(self._path(c, "regret"))
        if os.path.exists(self._path(c, "avg")):
            O.avg[self.nt:] = np.load(self._path(c, "avg"))
            O.avg_f64[self.first_board:] = np.load(self._path(c, "avg_f64"))
            if self.variant != "plus":
                O.avg_sum[self.nt:] = np.load(self._path(c, "avg_sum"))
        for p in (0, 1):
            if self.updated[p]:
                O.compute_new_strategy(p)  # a pure function of the regrets (CFRPlus.py:43-63)
        # the trunk's strategy (and dtype flags) come from the trunk instance
        O.strategy[:self.nt] = self.T.strategy[:self.nt]
        O.strat_f64[:self.first_board] = self.T.strat_f64[:self.first_board]
        return O

    def sweep(self, it, p):
        """evaluation of the current strategies (-> exploitability) and, if p is not None, seat p's half of iteration `it`"""
        groups = []
        for c in range(self.n_chunks):
            t0 = time.time()
            O = self._chunk_instance(c)
            O.update_reach()
            if self.pending is not None:  # Vanilla / Linear: seat q's half of iteration j ends here, with the reach just computed
                j, q = self.pending
                O.set_iter(j)
                O.add_strategy_to_average(q)
                np.save(self._path(c, "avg"), O.avg[self.nt:])
                np.save(self._path(c, "avg_f64"), O.avg_f64[self.first_board:])
                np.save(self._path(c, "avg_sum"), O.avg_sum[self.nt:])
            O.set_iter(it)
            O.compute_ev()
            roots = self.first_board + NB * np.arange(len(self.boards[c * self.chunk:(c + 1) * self.chunk]))
            vals = np.concatenate([O.ev[roots], O.ev_br[roots]], axis=1)  # [chunk][4][R]
            groups.append(group_sums(vals))
            if p is not None:
                O.compute_regrets(p)
                O.compute_new_strategy(p)
                np.save(self._path(c, "regret"), O.regret[self.nt:])
                if self.variant == "plus":  # CFR+'s average does not look at the reach: it is taken here
                    O.add_strategy_to_average(p)
                    np.save(self._path(c, "avg"), O.avg[self.nt:])
                    np.save(self._path(c, "avg_f64"), O.avg_f64[self.first_board:])
            print("  it %d seat %s chunk %d/%d  %.0f s" % (it, p, c + 1, self.n_chunks, time.time() - t0), flush=True)
        g = np.concatenate(groups)  # all groups in global order
        total = g[0].copy()
        for i in range(1, len(g)):
            total = total + g[i]
        T = self.T
        T.set_override(self.chance, total[0:2], total[2:4])
        T.set_iter(it)
        T.compute_ev()
        expl = np.array(T.exploitability, np.float32)
        self.pending = (it, p) if (p is not None and self.variant != "plus") else None
        if p is not None:
            T.compute_regrets(p)
            T.compute_new_strategy(p)
            T.update_reach()
            T.add_strategy_to_average(p)
            self.updated[p] = True
        return expl

    def run(self, n_iters):
        self.hist = []
        for it in range(n_iters):
            self.hist.append(self.sweep(it, 0))   # closes iteration it - 1 (or the reset) and does seat 0's half
            self.sweep(it, 1)
        self.hist.append(self.sweep(n_iters, None))
        return np.stack(self.hist)

    def eval_avg(self):
        """_CFRBase._evaluate_avg_strats (_CFRBase.py:218-262) chunk by chunk, after run(): every instance plays its AVERAGE strategy (float64, the trunk's
        from the trunk instance, the boards' from disk), reach, EVs, the canonical sum of the board roots into the trunk's chance node (symmetrised there
        for suit classes) -> exploitability of the average strategy; what orc_eval_avg does in one piece. Leaves the instances' strategies overwritten."""
        assert self.pending is None, "run() closes with an evaluation that applies the last pending average update"
        T, groups = self.T, []
        for c in range(self.n_chunks):
            O = self._chunk_instance(c)
            O.strategy[:self.nt] = T.avg[:self.nt]
            O.strat_f64[:self.first_board] = T.avg_f64[:self.first_board]
            O.strategy[self.nt:] = O.avg[self.nt:]
            O.strat_f64[self.first_board:] = O.avg_f64[self.first_board:]
            O.update_reach()
            O.compute_ev()
            roots = self.first_board + NB * np.arange(len(self.boards[c * self.chunk:(c + 1) * self.chunk]))
            groups.append(group_sums(np.concatenate([O.ev[roots], O.ev_br[roots]], axis=1)))
            print("  average-strategy evaluation chunk %d/%d" % (c + 1, self.n_chunks), flush=True)
        g = np.concatenate(groups)
        total = g[0].copy()
        for i in range(1, len(g)):
            total = total + g[i]
        T.strategy[:self.nt] = T.avg[:self.nt]
        T.strat_f64[:self.first_board] = T.avg_f64[:self.first_board]
        T.update_reach()
        T.set_override(self.chance, total[0:2], total[2:4])
        T.compute_ev()
        return np.array(T.exploitability, np.float32)

    def state_hashes(self):
        """sha-256 of regret / avg in the flat tree's column order: trunk columns, then every board's, chunk after chunk"""
        out = {}
        for name, arr in (("regret", self.T.regret), ("avg", self.T.avg)) + ((("avg_sum", self.T.avg_sum),) if self.variant != "plus" else ()):
            h = hashlib.sha256()
            fold = lambda a: np.ascontiguousarray(a + a.dtype.type(0)).tobytes()  # "+ 0" folds -0.0 into +0.0, as helpers.h32
            h.update(fold(np.asarray(arr[:self.nt])))
            for c in range(self.n_chunks):
                h.update(fold(np.load(self._path(c, name))))
            out[name] = h.hexdigest()
        return out

    def full_arrays(self):
        return {name: np.concatenate([np.asarray(getattr(self.T, name))[:self.nt]] + [np.load(self._path(c, name)) for c in range(self.n_chunks)])
                for name in ("regret", "avg") + (("avg_sum",) if self.variant != "plus" else ())}


def selftest(workdir, variant="plus"):
    boards = pc.fhp_boards(2048, seed=5)
    for f in os.listdir(workdir):
        if f.startswith("chunk"):
            os.remove(os.path.join(workdir, f))
    ch = Chunked(boards, 1024, workdir, variant)
    hist = ch.run(2)
    t, o = make(boards)
    o.cfr_reset(ch.vcode, 0)
    want = [np.array(o.exploitability, np.float32)]
    for _ in range(2):
        o.cfr_iteration()
        want.append(np.array(o.exploitability, np.float32))
    assert np.array_equal(hist, np.stack(want)), (hist, want)
    full = ch.full_arrays()
    assert np.array_equal(full["regret"], np.asarray(o.regret)) and np.array_equal(full["avg"], np.asarray(o.avg))
    hs = ch.state_hashes()
    assert hs["regret"] == h32(np.asarray(o.regret)) and hs["avg"] == h32(np.asarray(o.avg))
    if variant != "plus":
        assert np.array_equal(full["avg_sum"], np.asarray(o.avg_sum)) and hs["avg_sum"] == h32(np.asarray(o.avg_sum))
    ea = ch.eval_avg()
    assert np.array_equal(ea, o.eval_avg()), (ea, o.eval_avg())
    print("selftest ok: the chunked run equals the one-piece oracle (2048 boards, 2 iterations, %s; average-strategy exploitability %s)" % (variant, ea))


def selftest_weighted(workdir, variant="plus"):
    """weights + orbit-mean chance values + a ragged last chunk against the one-piece oracle: the first 2363 suit classes of Flop5Holdem (two full
    chunks of 1024 and one of 315 = 9 blocks of 32 + 27), 3 iterations"""
    from pokerrl_amd.game import board_enum
    reps, mult = pc.iso_classes(2363)
    cls = board_enum.hand_suit_classes(G.Flop5Holdem)
    for f in os.listdir(workdir):
        if f.startswith("chunk"):
            os.remove(os.path.join(workdir, f))
    ch = Chunked(reps, 1024, workdir, variant, mult=mult, sym_class=cls)
    hist = ch.run(3)
    t, o = make(reps)
    o.set_board_weights(mult)
    o.set_symmetrize(cls)
    o.cfr_reset(ch.vcode, 0)
    want = [np.array(o.exploitability, np.float32)]
    for _ in range(3):
        o.cfr_iteration()
        want.append(np.array(o.exploitability, np.float32))
    assert np.array_equal(hist, np.stack(want)), (hist, want)
    full = ch.full_arrays()
    assert np.array_equal(full["regret"], np.asarray(o.regret)) and np.array_equal(full["avg"], np.asarray(o.avg))
    hs = ch.state_hashes()
    assert hs["regret"] == h32(np.asarray(o.regret)) and hs["avg"] == h32(np.asarray(o.avg))
    if variant != "plus":
        assert np.array_equal(full["avg_sum"], np.asarray(o.avg_sum)) and hs["avg_sum"] == h32(np.asarray(o.avg_sum))
    ea = ch.eval_avg()
    assert np.array_equal(ea, o.eval_avg()), (ea, o.eval_avg())
    print("selftest ok: the chunked weighted run equals the one-piece oracle (2363 suit classes in chunks of 1024 + 1024 + 315, 3 iterations, %s)" % variant, hist[-1], "average:", ea)


def main_whole_game(n_iters, chunk, workdir, variant="plus"):
    """the WHOLE Flop5Holdem game: every board through its suit class (134 459 representatives x multiplicities 4 / 12 / 24 = 2 598 960 boards)"""
    from pokerrl_amd.game import board_enum
    reps, mult = board_enum.single_deal_board_classes(G.Flop5Holdem)
    assert int(mult.sum()) == 2598960 and len(reps) == 134459
    ch = Chunked(reps, chunk, workdir, variant, mult=mult, sym_class=board_enum.hand_suit_classes(G.Flop5Holdem))
    hist = ch.run(n_iters)
    hs = ch.state_hashes()
    ea = ch.eval_avg()  # (round 6) the average strategy's exploitability at full size: the <EVAL, AVG, AVG> pass over a ragged last chunk + orbit means
    out = os.path.join(HERE, "fhp_whole_game_%s_chunked.npz" % variant)
    np.savez(out, n_classes=len(reps), n_boards=int(mult.sum()), variant=variant, n_iters=n_iters, chunk=chunk, boards_sha256=h32(reps), mult_sha256=h32(mult.astype(np.int32)),
             expl_history=hist, regret_sha256=hs["regret"], avg_sha256=hs["avg"], avg_sum_sha256=hs.get("avg_sum", ""), eval_avg=ea, numpy=np.__version__)
    print("wrote", out, hist, "average strategy:", ea)


def main(n_boards, n_iters, chunk, workdir, variant="plus", seed=0):
    sys.path.insert(0, os.path.join(HERE, "..", ".."))
    import bench
    boards = bench.seeded_boards(n_boards, seed)  # bench.py's rank-0 board list
    ch = Chunked(boards, chunk, workdir, variant)
    hist = ch.run(n_iters)
    hs = ch.state_hashes()
    out = os.path.join(HERE, "fhp_%d_%s_chunked.npz" % (n_boards, variant))
    np.savez(out, n_boards=n_boards, seed=seed, variant=variant, n_iters=n_iters, chunk=chunk, boards_sha256=h32(boards), expl_history=hist,
             regret_sha256=hs["regret"], avg_sha256=hs["avg"], avg_sum_sha256=hs.get("avg_sum", ""), numpy=np.__version__)
    print("wrote", out, hist)


if __name__ == "__main__":
    a = sys.argv[1:]
    if a and a[0] == "--selftest":   # --selftest [workdir] [variant | weighted]
        os.makedirs(a[1] if len(a) > 1 else "/tmp/prl_chunked_selftest", exist_ok=True)
        if len(a) > 2 and a[2] == "weighted":   # --selftest <dir> weighted [variant]
            selftest_weighted(a[1], a[3] if len(a) > 3 else "plus")
        else:
            selftest(a[1] if len(a) > 1 else "/tmp/prl_chunked_selftest", a[2] if len(a) > 2 else "plus")
    elif a and a[0] == "--whole-game":   # --whole-game [n_iters] [chunk] [workdir] [variant]
        wd = a[3] if len(a) > 3 else "/tmp/prl_chunked_whole"
        os.makedirs(wd, exist_ok=True)
        main_whole_game(int(a[1]) if len(a) > 1 else 4, int(a[2]) if len(a) > 2 else 16384, wd, a[4] if len(a) > 4 else "plus")
    else:
        wd = a[3] if len(a) > 3 else "/tmp/prl_chunked"
        os.makedirs(wd, exist_ok=True)
        main(int(a[0]) if a else 262144, int(a[1]) if len(a) > 1 else 2, int(a[2]) if len(a) > 2 else 16384, wd, a[4] if len(a) > 4 else "plus")
