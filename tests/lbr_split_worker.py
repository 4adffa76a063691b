"""One rank of the LBR hand-split test (spawned by tests/test_lbr.py): BatchedLBR.run_sharded over gloo on the emulator build.
argv: lib_path out_dir n_hands_total seed [hash|table]   (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT from the env)
"table": every rank solves the agent's game itself (CFR+, deterministic: the ranks hold identical tables) and plays LBR against the average strategy."""
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    lib_path, out_dir = sys.argv[1:3]
    n_total, seed = int(sys.argv[3]), int(sys.argv[4])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world,
                                init_method="tcp://%s:%s" % (os.environ.get("MASTER_ADDR", "127.0.0.1"), os.environ["MASTER_PORT"]))
    from pokerrl_amd import _native
    L = _native.bind(lib_path)
    _native.lib = lambda: L
    _native.require_device = lambda: None
    import test_lbr as T
    game_cls, agent_bets, lbr_kwargs = T.CASES["DiscretizedNLLeduc"]
    t_prof = T.make_t_prof(game_cls, agent_bets, lbr_kwargs, n_total, tempfile.mkdtemp())
    if len(sys.argv) > 5 and sys.argv[5] == "table":
        table, _cfr = T.solved_table(game_cls, agent_bets, 4)
        b = T.BatchedLBR(t_prof, agent_kind="table", agent_seed=7, table=table)
    else:
        b = T.BatchedLBR(t_prof, agent_kind="hash", agent_seed=7)
    out = {}
    for seat in (0, 1):
        mean, conf, n, x = b.run_sharded(seat, n_total, deck_seed=seed + seat, device="cpu")
        out["mean%d" % seat], out["conf%d" % seat], out["n%d" % seat], out["x%d" % seat] = mean, conf, n, x
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **out)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
