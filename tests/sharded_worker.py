"""One rank of the sharded-solve tests (spawned by tests/test_sharded.py; also usable under torch.distributed.run).

argv: lib_path device out_dir n_local n_iters delay seed [total]   (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT from the env)
Solves the r-th contiguous block of fhp_boards(world * n_local, seed) through NativeSolver(shard=...) and writes the
rank's state to out_dir/rank<r>.npz. With `total` < world * n_local the shards are ragged: the last rank holds the rest."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    lib_path, device, out_dir = sys.argv[1:4]
    n_local, n_iters, delay, seed = (int(x) for x in sys.argv[4:8])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world,
                            init_method="tcp://%s:%s" % (os.environ.get("MASTER_ADDR", "127.0.0.1"), os.environ["MASTER_PORT"]))
    from pokerrl_amd import _native
    from pokerrl_amd.dist import TorchExchange
    from pokerrl_amd.game import bet_sets
    from pokerrl_amd.game import games as G
    import parity_cases as pc
    from helpers import env_args

    L = _native.bind(lib_path)
    total = int(sys.argv[8]) if len(sys.argv) > 8 else world * n_local
    boards = pc.fhp_boards(total, seed=seed, with_special=False)[rank * n_local:(rank + 1) * n_local]
    n_mine = len(boards)
    args = env_args(G.Flop5Holdem, 20000, bet_sets.POT_ONLY)
    t = _native.NativeTree(G.Flop5Holdem.native_game(args), G.Flop5Holdem.native_rules(), boards, _lib=L)
    ex = TorchExchange(device)
    s = _native.NativeSolver(t, "plus", delay, _lib=L, shard=(world, rank, ex) if total == world * n_local else (world, rank, ex, n_local, total))
    assert s.engine == "fused"
    s.iteration()
    s.iterations(n_iters - 1)
    out = dict(expl_history=s.get("expl_history"), regret=s.get("regret"), avg=s.get("avg"), eval_avg=s.eval_avg(),
               exchanges=np.int64(ex.calls), n_trunk_cols=np.int64(t.n_cols - n_mine * 14))
    # exact best response of an explicit strategy (BASELINE config 4: BR with the boards partitioned over the GPUs): every
    # rank loads the trunk columns + its own boards' columns of the same seeded strategy
    nt, per = int(t.n_cols - n_mine * 14), n_local * 14
    full = pc.seeded_strategy_for_sharding(nt, total, t.range_size, seed + 1)
    local = np.concatenate([full[:nt], full[nt + rank * per: nt + rank * per + n_mine * 14]])
    s.set_strategy(local)
    s.compute_ev()
    out["br_of_random"] = s.exploitability()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
