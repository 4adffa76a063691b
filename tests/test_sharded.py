"""Sharded solve: boards partitioned over ranks, one all-gather of canonical partial sums per EV pass.
The result must be bit-identical to the unsharded solve of the whole board list, for any world size.

CPU: world_size 2 over gloo, every rank drives the emulator build of the library (same kernel sources, same host code).
GPU: the product library, two processes sharing the one GPU of the test box (gloo + host staging instead of RCCL)."""
import os
import socket
import subprocess
import sys
import tempfile

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import parity_cases as pc  # noqa: E402
from helpers import env_args  # noqa: E402
from pokerrl_amd import _native  # noqa: E402
from pokerrl_amd.game import bet_sets  # noqa: E402
from pokerrl_amd.game import games as G  # noqa: E402


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_sharded(lib_path, device, world, n_local, n_iters, delay, seed, timeout, total=None):
    port = _free_port()
    with tempfile.TemporaryDirectory() as d:
        procs = []
        for r in range(world):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "sharded_worker.py"), lib_path, device, d,
                                           str(n_local), str(n_iters), str(delay), str(seed)] + ([str(total)] if total else []), env=env))
        try:
            for p in procs:
                assert p.wait(timeout=timeout) == 0
        finally:
            for p in procs:
                if p.poll() is None:
                    p.kill()
        return [dict(np.load(os.path.join(d, "rank%d.npz" % r))) for r in range(world)]


def check_against_union(L, ranks, world, n_local, n_iters, delay, seed, total=None):
    total = total or world * n_local
    boards = pc.fhp_boards(total, seed=seed, with_special=False)
    args = env_args(G.Flop5Holdem, 20000, bet_sets.POT_ONLY)
    t = _native.NativeTree(G.Flop5Holdem.native_game(args), G.Flop5Holdem.native_rules(), boards, _lib=L)
    s = _native.NativeSolver(t, "plus", delay, engine="fused", _lib=L)
    s.iterations(n_iters)
    hist, regret, avg, ev_avg = s.get("expl_history"), s.get("regret"), s.get("avg"), s.eval_avg()
    s.set_strategy(pc.seeded_strategy_for_sharding(t.n_cols - total * 14, total, t.range_size, seed + 1))
    s.compute_ev()
    br = s.exploitability()
    per = n_local * 14
    for r, out in enumerate(ranks):
        nt = int(out["n_trunk_cols"])
        assert np.array_equal(out["expl_history"], hist), "rank %d: exploitability history" % r
        assert np.array_equal(out["eval_avg"], ev_avg), "rank %d: average-strategy exploitability" % r
        assert np.array_equal(out["br_of_random"], br), "rank %d: best response of an explicit strategy" % r
        for name, full in (("regret", regret), ("avg", avg)):
            assert np.array_equal(out[name][:nt], full[:nt]), "rank %d: trunk %s" % (r, name)
            assert np.array_equal(out[name][nt:], full[nt + r * per: nt + min((r + 1) * n_local, total) * 14]), "rank %d: board %s" % (r, name)
        # one exchange per EV pass: reset (1), iteration() = 3, every further batched iteration 2, the batch's closing
        # evaluation 1, eval_avg 1
        assert int(out["exchanges"]) == 1 + 3 + (2 * (n_iters - 1) + 1 if n_iters > 1 else 0) + 1, out["exchanges"]


def chance_sum_reference(vals):
    """DESIGN.md canonical order in NumPy float32: blocks of 32 boards, groups of 32 blocks, then the groups, all sequential."""
    def level(x, fan):
        out = []
        for lo in range(0, len(x), fan):
            acc = x[lo].copy()
            for i in range(lo + 1, min(lo + fan, len(x))):
                acc = acc + x[i]
            out.append(acc)
        return out
    x = [v.astype(np.float32) for v in vals]
    x = level(level(x, 32), 32)
    return level(x, len(x))[0]


def check_chance_sum(L, n_boards, worlds):
    import ctypes
    rng = np.random.RandomState(n_boards)
    R = 6
    vals = (rng.random_sample((n_boards, 2, R)) * np.exp(rng.uniform(-8, 8, (n_boards, 2, R)))).astype(np.float32)
    want = chance_sum_reference(vals.reshape(n_boards, 2 * R)).reshape(2, R)
    for w in worlds:
        out = np.zeros((2, R), np.float32)
        rc = L.prl_chance_sum_host(vals.ctypes.data_as(ctypes.c_void_p), n_boards, R, w, out.ctypes.data_as(ctypes.c_void_p))
        assert rc == 0
        assert np.array_equal(out, want), "world %d" % w


def check_chance_sum_ragged(L, n_boards, world, shard_boards):
    """every rank before the last holds shard_boards boards, the last one the rest: same bits as the one-rank sum"""
    import ctypes
    rng = np.random.RandomState(n_boards + world)
    R = 6
    vals = (rng.random_sample((n_boards, 2, R)) * np.exp(rng.uniform(-8, 8, (n_boards, 2, R)))).astype(np.float32)
    want = chance_sum_reference(vals.reshape(n_boards, 2 * R)).reshape(2, R)
    out = np.zeros((2, R), np.float32)
    rc = L.prl_chance_sum_host_ragged(vals.ctypes.data_as(ctypes.c_void_p), n_boards, R, world, shard_boards, out.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0
    assert np.array_equal(out, want), (n_boards, world, shard_boards)


@pytest.fixture(scope="module")
def EMU():
    sys.path.insert(0, os.path.join(HERE, "emu"))
    import build_emu
    return build_emu.build()


def test_chance_sum_levels_do_not_depend_on_world_size_emu(EMU):
    L = _native.bind(EMU)
    check_chance_sum(L, 2048, (1, 2))       # shards of 1024 boards: whole groups are exchanged
    check_chance_sum(L, 2048 + 64, (1, 2, 3, 11))  # 1056 / 704 / 192 boards per rank: blocks
    check_chance_sum(L, 70, (1, 2, 5, 7))   # ragged: per-board values are exchanged
    check_chance_sum_ragged(L, 2048 + 48, 3, 1024)   # whole groups + a last shard of 48 boards (a partial group)
    check_chance_sum_ragged(L, 3 * 1024 + 1, 2, 2048)
    check_chance_sum_ragged(L, 200, 3, 96)           # blocks; the last shard is 8 boards
    check_chance_sum_ragged(L, 23, 4, 7)             # single boards
    out = np.zeros((2, 6), np.float32)  # a last shard that would be empty or longer than the others is refused
    import ctypes
    for n, w, sb in ((64, 3, 32), (100, 2, 40)):
        assert L.prl_chance_sum_host_ragged(out.ctypes.data_as(ctypes.c_void_p), n, 6, w, sb, out.ctypes.data_as(ctypes.c_void_p)) != 0


def test_sharded_ragged_world2_gloo_emu(EMU):
    """3 boards over 2 ranks (2 + 1): the all-boards configuration in small (`bench.py --all-boards`)"""
    ranks = run_sharded(EMU, "cpu", 2, 2, 2, 0, 33, timeout=900, total=3)
    check_against_union(_native.bind(EMU), ranks, 2, 2, 2, 0, 33, total=3)


def test_sharded_world2_gloo_emu(EMU):
    ranks = run_sharded(EMU, "cpu", 2, 2, 3, 1, 21, timeout=900)
    check_against_union(_native.bind(EMU), ranks, 2, 2, 3, 1, 21)


@pytest.mark.parametrize("game_args", [["--max-raises", "1,1,1,1"], ["--game", "DiscretizedNLHoldem", "--stack", "600"]])
def test_bench_multistreet_gpus_2_gloo_emu(EMU, game_args):
    """`python bench_multistreet.py --gpus 2`: two ranks, one flop each (with its turn / river run-outs), the per-street engine sharded over the
    first deal's outcomes; the same exploitability as the one-rank solve of both flops, ONE JSON line from rank 0. LimitHoldem with one raise per
    round, and DiscretizedNLHoldem with pot-sized raises (mixed street shapes, all-in run-out chains: round 6)"""
    import json
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["PRL_BENCH_EMU_LIB"] = EMU
    common = ["--turns", "2", "--rivers", "1", "--steps", "1", "--warmup", "1", "--no-cpu-baseline"] + game_args
    cmd = [sys.executable, os.path.join(root, "bench_multistreet.py")]
    two = subprocess.run(cmd + ["--gpus", "2", "--flops", "1"] + common, env=env, capture_output=True, text=True, timeout=900, check=True).stdout
    line = [x for x in two.splitlines() if x.startswith("{")]
    assert len(line) == 1, two
    j2 = json.loads(line[0])
    one = subprocess.run(cmd + ["--gpus", "1", "--flops", "2"] + common, env=env, capture_output=True, text=True, timeout=900, check=True).stdout
    j1 = json.loads([x for x in one.splitlines() if x.startswith("{")][0])
    assert j2["n_gpus"] == 2 and j1["n_gpus"] == 1 and j2["config"]["exchanges"] > 0 and j1["config"]["exchanges"] == 0
    assert j2["config"]["engine"].startswith("fused") and j2["build_flavor"].startswith("emu")
    assert j2["config"]["nodes_whole_job"] == j1["config"]["nodes_whole_job"] == j1["config"]["nodes"]
    assert j2["config"]["exploitability_chips"] == j1["config"]["exploitability_chips"]


def test_bench_gpus_2_launches_two_ranks_gloo_emu(EMU):
    """`python bench.py --gpus 2` with no launcher around it (the driver's form for N = 1; for N > 1 it wraps the same command in
    torch.distributed.run) must start two ranks itself, shard the boards, exchange once per EV pass and report n_gpus = 2. Here
    the ranks drive the emulator build over gloo; on the GPU box the same code path runs RCCL on the product library."""
    import json
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["PRL_BENCH_EMU_LIB"] = EMU
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--boards", "2", "--no-cpu-baseline"]
    out = subprocess.run(cmd + ["--fixed-check-boards", "3"], env=env, capture_output=True, text=True, timeout=900, check=True).stdout
    line = [x for x in out.splitlines() if x.startswith("{")]
    assert len(line) == 1, out  # ONE JSON line, printed by rank 0
    j = json.loads(line[0])
    assert j["n_gpus"] == 2 and j["steps"] == 1 and j["scaling"] == "weak"
    assert j["config"]["nodes_whole_tree"] == 5 + 15 * 4 and j["config"]["exchanges"] > 0 and j["config"]["iterations_done"] == 2
    # the two-rank solve of the 4 boards equals the one-rank solve of the same 4 boards
    one = subprocess.run(cmd[:2] + ["--gpus", "1", "--steps", "1", "--warmup", "1", "--boards", "4", "--no-cpu-baseline", "--fixed-check-boards", "3"], env=env,
                         capture_output=True, text=True, timeout=900, check=True).stdout
    j1 = json.loads([x for x in one.splitlines() if x.startswith("{")][0])
    assert j1["n_gpus"] == 1 and j1["config"]["exchanges"] == 0
    assert j1["config"]["exploitability_mbb_per_g"] == j["config"]["exploitability_mbb_per_g"]
    # the fixed problem the SCALE lines carry (one list of boards whatever the world size): the same bits from one rank and from two
    f1, f2 = j1["config"]["fixed_problem_check"], j["config"]["fixed_problem_check"]
    assert (f1["world"], f2["world"], f1["total_boards"], f2["total_boards"]) == (1, 2, 3, 3)
    assert f1["exploitability_f32_hex"] == f2["exploitability_f32_hex"] and f1["avg_strategy_exploitability_f32_hex"] == f2["avg_strategy_exploitability_f32_hex"]
    # ONE list of 3 boards over two ranks (2 + 1, `--all-boards` in small): strong scaling, same result as the one-rank solve of the 3
    rag = subprocess.run(cmd[:2] + ["--gpus", "2", "--steps", "1", "--warmup", "1", "--total-boards", "3", "--no-cpu-baseline"], env=env,
                         capture_output=True, text=True, timeout=900, check=True).stdout
    j2 = json.loads([x for x in rag.splitlines() if x.startswith("{")][0])
    one3 = subprocess.run(cmd[:2] + ["--gpus", "1", "--steps", "1", "--warmup", "1", "--boards", "3", "--no-cpu-baseline"], env=env,
                          capture_output=True, text=True, timeout=900, check=True).stdout
    j3 = json.loads([x for x in one3.splitlines() if x.startswith("{")][0])
    assert j2["n_gpus"] == 2 and j2["scaling"] == "strong" and j2["config"]["nodes_whole_tree"] == 5 + 15 * 3 == j3["config"]["nodes_whole_tree"]
    assert j2["config"]["exploitability_mbb_per_g"] == j3["config"]["exploitability_mbb_per_g"]
    assert j2["config"]["avg_strategy_exploitability_mbb_per_g"] == j3["config"]["avg_strategy_exploitability_mbb_per_g"]


def test_bench_br_gpus_2_launches_two_ranks_gloo_emu(EMU):
    """`python bench_br.py --gpus 2` (BASELINE config 4: exact best response with the boards sharded over the GPUs) starts its two ranks,
    shards the boards and reports n_gpus = 2 with a non-zero exchange count; here over gloo on the emulator build"""
    import json
    root = os.path.dirname(HERE)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["PRL_BENCH_EMU_LIB"] = EMU
    cmd = [sys.executable, os.path.join(root, "bench_br.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--boards", "2", "--no-cpu-baseline"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, check=True).stdout
    line = [x for x in out.splitlines() if x.startswith("{")]
    assert len(line) == 1, out
    j = json.loads(line[0])
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["config"]["exchanges"] > 0
    assert j["config"]["nodes_whole_tree"] == 5 + 15 * 4 and j["config"]["engine"] == "fused"
    assert j["roofline"]["kernel"].startswith("prl_k_fhp_pass") and j["config"]["exploitability_mbb_per_g"] > 0


def _preflight_ranks(bad_rank, id_fails, extra_env=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env or {})
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(HERE, "rccl_preflight_worker.py"), str(bad_rank), "1" if id_fails else "0"]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)  # a hang (one rank inside a collective) is the failure mode
    assert out.returncode == 0, out.stderr[-2000:]
    return [x for x in out.stdout.splitlines() if x.startswith(("RAISED", "ID"))]


@pytest.mark.parametrize("bad_rank", [0, 1])
def test_rccl_shard_preflight_one_rank_cannot_bind_rccl_every_rank_raises(bad_rank):
    """pokerrl_amd.dist.rccl_shard: a rank that cannot bind RCCL (rank 0 OR rank 1: e.g. a mistyped PRL_RCCL_LIB on one host) makes EVERY rank raise
    before anybody enters ncclCommInitRank -- so bench.py's agreement (all ranks fall back to --exchange torch) is reached instead of a hang"""
    lines = _preflight_ranks(bad_rank, False)
    assert len(lines) == 2 and all(x.startswith("RAISED") and "every rank raises" in x for x in lines), lines


def test_rccl_shard_preflight_forced_failure_on_rank_1_and_id_failure_on_rank_0():
    lines = _preflight_ranks(-1, False, {"PRL_TEST_RCCL_FAIL_RANK": "1"})  # both can bind, the environment knob fails rank 1
    assert len(lines) == 2 and all(x.startswith("RAISED") for x in lines), lines
    lines = _preflight_ranks(-1, True)  # everybody can bind, rank 0 cannot draw the communicator id: the status byte of the broadcast
    assert len(lines) == 2 and all(x.startswith("RAISED") and "could not draw" in x for x in lines), lines
    lines = _preflight_ranks(-1, False)  # nothing fails: both ranks hold the same 128 bytes
    assert len(lines) == 2 and all(x.startswith("ID") for x in lines) and lines[0] == lines[1], lines


def test_bench_shard_geometry():
    import bench
    for total, world in ((2598960, 8), (2598960, 4), (3, 2), (100, 3), (5000, 2), (7, 1), (2048, 2)):
        g = [bench.shard_geometry(total, world, r) for r in range(world)]
        assert sum(x[1] for x in g) == total and all(0 < x[1] <= x[0] for x in g) and len({x[0] for x in g}) == 1
    assert bench.shard_geometry(2598960, 8, 0) == (325632, 325632) and bench.shard_geometry(2598960, 8, 7) == (325632, 319536)


# ---- multi-street trees on the per-street fused engine: the flops (first-deal outcomes) are what the ranks split -----------------------
def _top_blocks(t):
    """column ranges of a multi-street flat tree: (trunk columns, {(trunk chance node index, outcome k): columns of that subtree}), DFS order"""
    kind, sub, fc, nch = t.field("kind"), t.field("subtree_size"), t.field("first_col"), t.field("n_children")
    cs, cl = t.field("child_start"), t.field("child_list")
    ncols_below = lambda n: int(np.sum(nch[n:n + sub[n]][kind[n:n + sub[n]] == 0]))
    trunk, blocks, n, j = [], {}, 0, 0
    while n < t.n_nodes:
        if kind[n] == 1:
            for k in range(nch[n]):
                root = int(cl[cs[n] + k])
                blocks[(j, k)] = np.arange(fc[root], fc[root] + ncols_below(root))
            j += 1
            n += int(sub[n])
            continue
        if kind[n] == 0:
            trunk.extend(range(fc[n], fc[n] + nch[n]))
        n += 1
    return np.array(trunk, np.int64), blocks


def run_sharded_streets(lib_path, device, world, cfg, timeout):
    import json
    port = _free_port()
    with tempfile.TemporaryDirectory() as d:
        procs = []
        for r in range(world):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "sharded_streets_worker.py"), lib_path, device, d, json.dumps(cfg)], env=env))
        try:
            for p in procs:
                assert p.wait(timeout=timeout) == 0
        finally:
            for p in procs:
                if p.poll() is None:
                    p.kill()
        return [dict(np.load(os.path.join(d, "rank%d.npz" % r))) for r in range(world)]


def check_streets_against_union(L, ranks, world, cfg):
    """every rank's state = its flops' part of the one-rank solve of all flops, bit for bit (regrets, averages: the trunk's columns
    and every street instance's; exploitability history; average-strategy exploitability)"""
    n_local, per_flop = cfg["n_local"], cfg["n_turns"] * cfg["n_rivers"]
    runouts = pc.multistreet_runouts(world * n_local, cfg["n_turns"], cfg["n_rivers"], seed=cfg["seed"])
    game_cls = getattr(G, cfg.get("game", "LimitHoldem"))
    game = game_cls.native_game(env_args(game_cls, cfg.get("stack", 48), cfg.get("bets")))
    for i, v in enumerate(cfg.get("max_raises") or []):
        game.max_raises[i] = v
    t = _native.NativeTree(game, game_cls.native_rules(), runouts, _lib=L)
    s = _native.NativeSolver(t, cfg.get("variant", "plus"), 0, engine="fused", _lib=L)
    s.iterations(cfg["n_iters"])
    hist, regret, avg, ev_avg = s.get("expl_history"), s.get("regret"), s.get("avg"), s.eval_avg()
    trunk_u, blocks_u = _top_blocks(t)
    for r, out in enumerate(ranks):
        tl = _native.NativeTree(game, game_cls.native_rules(), runouts[r * n_local * per_flop:(r + 1) * n_local * per_flop], _lib=L)
        trunk_l, blocks_l = _top_blocks(tl)
        assert np.array_equal(out["expl_history"], hist), "rank %d: exploitability history" % r
        assert np.array_equal(out["eval_avg"], ev_avg), "rank %d: average-strategy exploitability" % r
        for name, full in (("regret", regret), ("avg", avg)):
            assert np.array_equal(out[name][trunk_l], full[trunk_u]), "rank %d: trunk %s" % (r, name)
            for (j, k), cols in blocks_l.items():
                assert np.array_equal(out[name][cols], full[blocks_u[(j, r * n_local + k)]]), "rank %d: %s of flop %d below trunk leaf %d" % (r, name, k, j)
        assert int(out["exchanges"]) == 1 + 3 + (2 * (cfg["n_iters"] - 1) + 1 if cfg["n_iters"] > 1 else 0) + 1, out["exchanges"]


def test_streets_sharded_world2_gloo_emu(EMU):
    """LimitHoldem (one raise per round), 2 flops x 2 turns x 1 river, one flop per rank: the sharded per-street solve over gloo on the
    emulator build equals the one-rank solve of both flops"""
    cfg = dict(n_local=1, n_turns=2, n_rivers=1, n_iters=2, seed=13, max_raises=[1, 1, 1, 1])
    ranks = run_sharded_streets(EMU, "cpu", 2, cfg, timeout=1500)
    check_streets_against_union(_native.bind(EMU), ranks, 2, cfg)


@pytest.mark.parametrize("game,stack,bets", [("LimitHoldem", 6, None), ("DiscretizedNLHoldem", 600, [1.0])])
def test_streets_sharded_mixed_streets_world2_gloo_emu(EMU, game, stack, bets):
    """MIXED STREETS sharded (csrc/prl_st.h): short-stacked LimitHoldem / DiscretizedNLHoldem with pot-sized raises -- several street shapes per street
    and all-in run-out chains, some of them directly below the trunk (rows of the first street that the ranks exchange like any instance's) --
    2 flops x 2 turns x 1 river, one flop per rank, equals the one-rank solve of both flops"""
    cfg = dict(n_local=1, n_turns=2, n_rivers=1, n_iters=2, seed=13, game=game, stack=stack, bets=bets, variant="linear" if bets else "plus")
    ranks = run_sharded_streets(EMU, "cpu", 2, cfg, timeout=1500)
    check_streets_against_union(_native.bind(EMU), ranks, 2, cfg)


@pytest.mark.gpu
@pytest.mark.parametrize("n_local,max_raises", [(1, None), (32, [1, 1, 1, 1])])  # flops / blocks of 32 flops are exchanged
def test_gpu_streets_sharded_world2_matches_unsharded(n_local, max_raises):
    """two processes on the one GPU: LimitHoldem with its full betting (27-node street subtrees) one flop per rank; a 9-node betting tree
    with 32 flops per rank (whole 32-flop blocks are what the ranks exchange)"""
    cfg = dict(n_local=n_local, n_turns=2, n_rivers=2 if n_local == 1 else 1, n_iters=3, seed=17, max_raises=max_raises)
    ranks = run_sharded_streets(_native.LIB_PATH, "cuda", 2, cfg, timeout=900)
    check_streets_against_union(_native.lib(), ranks, 2, cfg)


@pytest.mark.gpu
@pytest.mark.parametrize("game,stack,bets,n_local", [("DiscretizedNLHoldem", 2500, [1.0], 2), ("LimitHoldem", 10, None, 32)])
def test_gpu_streets_sharded_mixed_streets_world2_matches_unsharded(game, stack, bets, n_local):
    """two processes on the one GPU, mixed street shapes and all-in run-out chains (DiscretizedNLHoldem with pot-sized raises; LimitHoldem with 10-chip
    stacks, 32 flops per rank = whole blocks are exchanged): every rank's state = its part of the one-rank solve"""
    cfg = dict(n_local=n_local, n_turns=2, n_rivers=2 if n_local < 32 else 1, n_iters=3, seed=19, game=game, stack=stack, bets=bets, variant="plus")
    ranks = run_sharded_streets(_native.LIB_PATH, "cuda", 2, cfg, timeout=900)
    check_streets_against_union(_native.lib(), ranks, 2, cfg)


@pytest.mark.gpu
def test_gpu_chance_sum_levels_do_not_depend_on_world_size():
    L = _native.lib()
    check_chance_sum(L, 8192, (1, 2, 4, 8))
    check_chance_sum(L, 2048 + 64, (1, 2, 3, 11))
    check_chance_sum(L, 70, (1, 2, 5, 7))


@pytest.mark.gpu
@pytest.mark.parametrize("n_local,block_sum", [(3, None), (32, None), (32, "1"), (1024, None), (1024, "1")])  # boards / blocks / groups are exchanged
def test_gpu_sharded_world2_matches_unsharded(n_local, block_sum, monkeypatch):
    if block_sum:  # the pass sums its 32-board blocks itself (the default at >= ~32 boards per workgroup; forced here on small shards)
        monkeypatch.setenv("PRL_FHP_BLOCK_SUM", block_sum)
    ranks = run_sharded(_native.LIB_PATH, "cuda", 2, n_local, 4, 0, 33, timeout=600)
    check_against_union(_native.lib(), ranks, 2, n_local, 4, 0, 33)


@pytest.mark.gpu
@pytest.mark.parametrize("n_local,total", [(1024, 1024 + 48), (64, 100), (5, 8)])  # groups / blocks / boards, shorter last shard
def test_gpu_sharded_ragged_world2_matches_unsharded(n_local, total):
    ranks = run_sharded(_native.LIB_PATH, "cuda", 2, n_local, 3, 0, 34, timeout=600, total=total)
    check_against_union(_native.lib(), ranks, 2, n_local, 3, 0, 34, total=total)


@pytest.mark.gpu
def test_gpu_two_processes_at_bench_size_on_shuffled_backing():
    """bench.py's per-GPU shard (262144 boards, ~58 GB) twice on the one GPU of the test box, as the two ranks of a sharded solve: every rank's
    large arrays sit on the shuffled 2 MB virtual-memory backing a multi-GPU run uses (the test checks that they do), the all-gather goes over gloo
    through host staging. Afterwards the one-rank solve of all 524288 boards (116 GB) must show the same exploitability history and the same bits in
    the regret / average columns of each rank's first and last 512 boards."""
    import bench
    import sharded_vmm_worker as W
    n_local, n_iters, seed, world = 262144, 3, 0, 2
    port = _free_port()
    with tempfile.TemporaryDirectory() as d:
        procs = []
        for r in range(world):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
            env.pop("PRL_VMM_SHUFFLE_MB", None)
            procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "sharded_vmm_worker.py"), d, str(n_local), str(n_iters), str(seed)], env=env))
        try:
            for p in procs:
                assert p.wait(timeout=900) == 0
        finally:
            for p in procs:
                if p.poll() is None:
                    p.kill()
        ranks = [dict(np.load(os.path.join(d, "rank%d.npz" % r))) for r in range(world)]
    for out in ranks:
        n_arrays, n_bytes = (int(x) for x in out["vmm_ranges"])
        assert n_arrays >= 3 and n_bytes > 0.8 * int(out["bytes_allocated"][0]) > 40e9, (out["vmm_ranges"], out["bytes_allocated"])
    t = bench.fhp_tree(bench.seeded_boards(world * n_local, seed))
    s = _native.NativeSolver(t, "plus", 0, engine="fused")
    assert int(s.get("vmm_ranges")[0]) == 0  # an unsharded solve allocates plainly
    s.iterations(n_iters)
    hist, nt = s.get("expl_history"), int(t.n_cols - world * n_local * 14)
    k = W.SLICE_BOARDS
    for r, out in enumerate(ranks):
        assert np.array_equal(out["expl_history"], hist), "rank %d: exploitability history" % r
        assert list(out["first"]) == W.digests(s, nt, r * n_local, k), "rank %d: first boards" % r
        assert list(out["last"]) == W.digests(s, nt, (r + 1) * n_local - k, k), "rank %d: last boards" % r


@pytest.mark.gpu
def test_gpu_chance_sum_ragged_shards():
    L = _native.lib()
    check_chance_sum_ragged(L, 7 * 2048 + 48, 8, 2048)   # the all-boards geometry in small: whole groups, a short last shard
    check_chance_sum_ragged(L, 200, 3, 96)
    check_chance_sum_ragged(L, 23, 4, 7)


@pytest.mark.gpu
@pytest.mark.parametrize("how", ["rccl", "torch"])
def test_gpu_bench_exchange_path_over_rccl_one_rank(how):
    """bench.py with the all-gather path forced on, world_size 1 -- the same code the multi-GPU launch runs; its exploitability must
    equal the plain run's. rccl: the library's own exchange (prl_solver_create_sharded_rccl: ncclCommInitRank + ncclAllGather on the
    solver's stream, the default of a multi-GPU run); torch: torch.distributed nccl (= RCCL) on zero-copy views of the solver's device
    buffers through the C ABI's callback."""
    import json
    root = os.path.dirname(HERE)
    base = [sys.executable, os.path.join(root, "bench.py"), "--steps", "3", "--warmup", "1", "--boards", "2048", "--no-cpu-baseline", "--no-placement-probe"]
    env = dict(os.environ, MASTER_PORT=str(_free_port()))
    a = json.loads(subprocess.run(base, env=env, capture_output=True, text=True, timeout=600, check=True).stdout.strip().splitlines()[-1])
    b = json.loads(subprocess.run(base + ["--exchange", how], env=dict(env, PRL_BENCH_FORCE_EXCHANGE="1"), capture_output=True, text=True, timeout=600,
                                  check=True).stdout.strip().splitlines()[-1])
    assert b["config"]["exchanges"] > 0 and a["config"]["exchanges"] == 0 and b["config"]["exchange"] == how
    assert a["config"]["exploitability_mbb_per_g"] == b["config"]["exploitability_mbb_per_g"]
    assert a["config"]["iterations_done"] == b["config"]["iterations_done"] == 4


@pytest.mark.gpu
def test_gpu_bench_multistreet_exchange_path_over_rccl_one_rank_mixed_streets():
    """bench_multistreet.py --game DiscretizedNLHoldem with the library's own exchange forced on, world_size 1 (ncclAllGather on the solver's stream while
    the run-out forest runs on its own stream): the same exploitability as the plain run"""
    import json
    root = os.path.dirname(HERE)
    base = [sys.executable, os.path.join(root, "bench_multistreet.py"), "--game", "DiscretizedNLHoldem", "--flops", "4", "--turns", "2", "--rivers", "2", "--steps", "3",
            "--warmup", "1", "--no-cpu-baseline", "--placement-candidates", "1"]
    env = dict(os.environ, MASTER_PORT=str(_free_port()))
    def line(extra_env):
        r = subprocess.run(base, env=dict(env, **extra_env), capture_output=True, text=True, timeout=600)
        js = [x for x in r.stdout.splitlines() if x.startswith("{")]
        assert r.returncode == 0 and len(js) == 1, (r.returncode, r.stdout[-2000:], r.stderr[-3000:])
        return json.loads(js[0])
    a, b = line({}), line({"PRL_BENCH_FORCE_EXCHANGE": "1"})
    assert b["config"]["exchanges"] > 0 and a["config"]["exchanges"] == 0
    assert a["config"]["engine"].startswith("fused") and b["config"]["engine"].startswith("fused")
    assert a["config"]["exploitability_chips"] == b["config"]["exploitability_chips"] and a["config"]["iterations_done"] == b["config"]["iterations_done"]


@pytest.mark.gpu
def test_gpu_rccl_world2_on_one_gpu_or_the_reason_it_cannot_run():
    """Two ranks of the library's own RCCL exchange (shuffled virtual-memory backing on, as every sharded solve has it) on the ONE GPU of
    the test box. RCCL refuses two ranks of a communicator on one device ("Duplicate GPU detected"); when it does, the test is skipped with
    RCCL's own words on the record -- the multi-GPU run of this path is the driver's bench.py --gpus N."""
    port = _free_port()
    with tempfile.TemporaryDirectory() as d:
        procs = []
        for r in range(2):
            env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), NCCL_DEBUG="WARN", PRL_VMM_SHUFFLE_MB="2")
            procs.append(subprocess.Popen([sys.executable, os.path.join(HERE, "rccl_world2_worker.py"), d], env=env, stdout=subprocess.PIPE,
                                          stderr=subprocess.STDOUT, text=True))
        outs = []
        for p in procs:
            try:
                outs.append(p.communicate(timeout=150)[0])
            except subprocess.TimeoutExpired:
                p.kill()
                outs.append(p.communicate()[0] + "\n[timeout]")
        if all(p.returncode == 0 for p in procs):
            ranks = [dict(np.load(os.path.join(d, "rank%d.npz" % r))) for r in range(2)]
            assert np.array_equal(ranks[0]["expl_history"], ranks[1]["expl_history"])
            assert np.array_equal(ranks[0]["expl_history"], ranks[0]["want"])  # = the one-rank solve of both shards
            return
        reason = [ln for o in outs for ln in o.splitlines() if "uplicate" in ln] + [ln for o in outs for ln in o.splitlines() if "ncclCommInitRank" in ln] + \
                 [ln for o in outs for ln in o.splitlines() if "NCCL WARN" in ln and "iommu" not in ln]
        print("\n".join(reason[:6]))
        pytest.skip("RCCL does not run two ranks on one device: " + (reason[0] if reason else outs[0][-300:]))
