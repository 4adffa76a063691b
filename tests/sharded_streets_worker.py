"""One rank of the sharded multi-street tests (spawned by tests/test_sharded.py).

argv: lib_path device out_dir json   (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT from the env)
json: {"n_local": flops per rank, "n_turns":, "n_rivers":, "n_iters":, "seed":, "max_raises": [..] or null, "variant":, "game": (LimitHoldem), "stack": (48), "bets": (null)}
Rank r solves the r-th block of n_local flops (with all their turn / river run-outs) of pc.multistreet_runouts(world * n_local, ...) on the
per-street fused engine through NativeSolver(shard=...) and writes its state to out_dir/rank<r>.npz."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    lib_path, device, out_dir = sys.argv[1:4]
    cfg = json.loads(sys.argv[4])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world,
                            init_method="tcp://%s:%s" % (os.environ.get("MASTER_ADDR", "127.0.0.1"), os.environ["MASTER_PORT"]))
    from pokerrl_amd import _native
    from pokerrl_amd.dist import TorchExchange
    from pokerrl_amd.game import games as G
    import parity_cases as pc
    from helpers import env_args

    L = _native.bind(lib_path)
    n_local, per_flop = cfg["n_local"], cfg["n_turns"] * cfg["n_rivers"]
    runouts = pc.multistreet_runouts(world * n_local, cfg["n_turns"], cfg["n_rivers"], seed=cfg["seed"])
    mine = runouts[rank * n_local * per_flop:(rank + 1) * n_local * per_flop]
    game_cls = getattr(G, cfg.get("game", "LimitHoldem"))
    game = game_cls.native_game(env_args(game_cls, cfg.get("stack", 48), cfg.get("bets")))
    if cfg.get("max_raises"):
        for i, v in enumerate(cfg["max_raises"]):
            game.max_raises[i] = v
    t = _native.NativeTree(game, game_cls.native_rules(), mine, _lib=L)
    ex = TorchExchange(device)
    s = _native.NativeSolver(t, cfg.get("variant", "plus"), 0, _lib=L, shard=(world, rank, ex))
    assert s.engine == "fused"
    s.iteration()
    s.iterations(cfg["n_iters"] - 1)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), expl_history=s.get("expl_history"), regret=s.get("regret"), avg=s.get("avg"),
             eval_avg=s.eval_avg(), exchanges=np.int64(ex.calls))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
