"""Worker of test_rccl_shard_preflight_*: one rank of a gloo group calls pokerrl_amd.dist.rccl_shard with a FAKE library object (the agreement
logic is Python: which rank fails is the test's choice). Prints RAISED <message> or ID <hex of the 128 bytes it got>."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.distributed as dist  # noqa: E402

from pokerrl_amd.dist import rccl_shard  # noqa: E402


class FakeLib:
    def __init__(self, rank, bad_rank, id_fails):
        self.rank, self.bad_rank, self.id_fails = rank, bad_rank, id_fails

    def prl_rccl_info(self, buf, n):
        msg = b"PRL_RCCL_LIB=/nowhere/librccl.so: cannot open shared object file" if self.rank == self.bad_rank else b"/opt/rocm/lib/librccl.so.1"
        ctypes.memmove(buf, msg + b"\0", len(msg) + 1)
        return -5 if self.rank == self.bad_rank else 0

    def prl_rccl_unique_id(self, p):
        if self.id_fails:
            return -5
        ctypes.memmove(p, bytes(range(128)), 128)
        return 0

    def prl_last_error(self):
        return b"ncclGetUniqueId failed (fake)"


if __name__ == "__main__":
    bad_rank, id_fails = int(sys.argv[1]), sys.argv[2] == "1"
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    try:
        shard = rccl_shard(world, rank, lib=FakeLib(rank, bad_rank, id_fails))
        line = "ID " + bytes(shard[3]).hex()[:16]
    except RuntimeError as e:
        line = "RAISED " + str(e)[:200].replace("\n", " ")
    os.write(1, (line + "\n").encode())  # ONE write per rank: the ranks share the pipe
    dist.barrier()
    dist.destroy_process_group()
