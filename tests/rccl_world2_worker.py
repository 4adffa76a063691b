"""One rank of test_gpu_rccl_world2_on_one_gpu_or_the_reason_it_cannot_run: both ranks on device 0, the library's own RCCL exchange.
argv: out_dir   (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT from the env). Exits non-zero with RCCL's message if the communicator
cannot be formed (two ranks on one device)."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    out_dir = sys.argv[1]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)  # for the id broadcast only
    from pokerrl_amd import _native
    from pokerrl_amd.dist import rccl_shard
    from pokerrl_amd.game import bet_sets
    from pokerrl_amd.game import games as G
    import parity_cases as pc
    from helpers import env_args
    _native.require_device()
    n_local = 32
    boards = pc.fhp_boards(world * n_local, seed=41, with_special=False)
    args = env_args(G.Flop5Holdem, 20000, bet_sets.POT_ONLY)
    t = _native.NativeTree(G.Flop5Holdem.native_game(args), G.Flop5Holdem.native_rules(), boards[rank * n_local:(rank + 1) * n_local])
    try:
        s = _native.NativeSolver(t, "plus", 0, shard=rccl_shard(world, rank))
    except _native.NativeError as e:
        print("rank %d: %s" % (rank, e), flush=True)
        sys.exit(3)
    s.iterations(3)
    hist = s.get("expl_history")
    tu = _native.NativeTree(G.Flop5Holdem.native_game(args), G.Flop5Holdem.native_rules(), boards)
    u = _native.NativeSolver(tu, "plus", 0, engine="fused")
    u.iterations(3)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), expl_history=hist, want=u.get("expl_history"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
