"""One rank of the bench-size two-process test (tests/test_sharded.py::test_gpu_two_processes_at_bench_size_on_shuffled_backing).

argv: out_dir n_local n_iters seed   (RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT from the env; the product library, device cuda)
The rank's shard is bench.py's: boards [rank * n_local, (rank + 1) * n_local) of the seeded permutation of all C(52,5) boards. A sharded solve
backs its large arrays with shuffled 2 MB physical chunks under one virtual range; the rank reports how many arrays and bytes
that were, its exploitability history, and SHA-256 digests of the regret / average columns of its first and last 512 boards."""
import hashlib
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

import numpy as np  # noqa: E402
import torch.distributed as dist  # noqa: E402

SLICE_BOARDS = 512


def digests(s, n_trunk_cols, board0, n_boards):
    """SHA-256 of the regret / average columns of boards [board0, board0 + n_boards) of a solver whose board columns start at n_trunk_cols"""
    out = []
    for name in ("regret", "avg"):
        a = s.get_cols(name, n_trunk_cols + board0 * 14, n_boards * 14)
        out.append(hashlib.sha256(np.ascontiguousarray(a + a.dtype.type(0)).tobytes()).hexdigest())
    return out


def main():
    out_dir = sys.argv[1]
    n_local, n_iters, seed = (int(x) for x in sys.argv[2:5])
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world,
                            init_method="tcp://%s:%s" % (os.environ.get("MASTER_ADDR", "127.0.0.1"), os.environ["MASTER_PORT"]))
    import bench
    from pokerrl_amd import _native
    from pokerrl_amd.dist import TorchExchange

    t = bench.fhp_tree(bench.seeded_boards(n_local, seed, offset=rank * n_local))
    ex = TorchExchange("cuda")
    s = _native.NativeSolver(t, "plus", 0, shard=(world, rank, ex))
    assert s.engine == "fused"
    s.iterations(n_iters)
    nt = int(t.n_cols - n_local * 14)
    k = min(SLICE_BOARDS, n_local)
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), expl_history=s.get("expl_history"), vmm_ranges=s.get("vmm_ranges"),
             bytes_allocated=s.get("bytes_allocated"), first=np.array(digests(s, nt, 0, k)), last=np.array(digests(s, nt, n_local - k, k)),
             exchanges=np.int64(ex.calls))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
