"""
torch.distributed plumbing of the sharded solve (one process per GPU; backend "nccl" is RCCL over xGMI on ROCm).

The solver's C-ABI takes an exchange callback (include/pokerrl_hip.h: prl_exchange_fn) that all-gathers one device buffer
per EV pass: the chance node's partial sums of every rank's boards. TorchExchange wraps the raw
pointers as tensors without copying and runs `all_gather_into_tensor` on them. PyTorch is used for the collective only.
"""
import ctypes
import time

import numpy as np
import torch
import torch.distributed as dist


class _DevBuf:
    """Zero-copy view of a raw device pointer for torch.as_tensor (CUDA array interface v2)."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


class TorchExchange:
    """exchange(local_ptr, gathered_ptr, bytes_per_rank) for NativeSolver(shard=...).

    device="cuda": the pointers are HBM addresses; with the nccl backend the collective runs on them directly, with any
                   other backend (gloo: tests on a single GPU) it is staged through host memory.
    device="cpu":  the pointers are host addresses (the CPU test-suite's emulator build of the library)."""

    def __init__(self, device, group=None):
        self.device = device
        self.group = group
        self.world = dist.get_world_size(group)
        self.calls = 0
        self.bytes = 0
        self.seconds = 0.0
        self._views = {}  # the solver's exchange buffers never move: wrap each (pointer, size) once
        self._stream = None

    def bind_stream(self, hip_stream_ptr):
        """Called by NativeSolver with its HIP stream. With the nccl backend the collective is enqueued under that stream
        (torch.cuda.ExternalStream): ProcessGroupNCCL orders it after the stream's pending kernels and makes the stream wait
        for its completion, so neither side blocks the host. Returns True if the exchange is now stream-ordered. The reset
        at solver creation already ran one exchange the synchronous way."""
        if self.device != "cuda" or dist.get_backend(self.group) != "nccl" or not hip_stream_ptr:
            return False
        self._stream = torch.cuda.ExternalStream(hip_stream_ptr)
        return True

    def _view(self, ptr, nbytes):
        key = (ptr, nbytes)
        v = self._views.get(key)
        if v is None:
            v = self._views[key] = self._make_view(ptr, nbytes)
        return v

    def _make_view(self, ptr, nbytes):
        if self.device == "cpu":
            return torch.from_numpy(np.ctypeslib.as_array((ctypes.c_uint8 * nbytes).from_address(ptr)))
        return torch.as_tensor(_DevBuf(ptr, nbytes), device="cuda")

    def __call__(self, local_ptr, gathered_ptr, nbytes):
        t0 = time.perf_counter()
        local = self._view(local_ptr, nbytes)
        gathered = self._view(gathered_ptr, nbytes * self.world)
        if self.device == "cuda" and dist.get_backend(self.group) != "nccl":
            host = torch.empty(nbytes * self.world, dtype=torch.uint8)
            dist.all_gather_into_tensor(host, local.cpu(), group=self.group)
            gathered.copy_(host)
        elif self._stream is not None:
            with torch.cuda.stream(self._stream):
                dist.all_gather_into_tensor(gathered, local, group=self.group)
        else:
            dist.all_gather_into_tensor(gathered, local, group=self.group)
        if self.device == "cuda" and self._stream is None:
            torch.cuda.synchronize()  # the solver's own stream continues only after the gathered buffer is complete
        self.calls += 1
        self.bytes += nbytes * self.world
        self.seconds += time.perf_counter() - t0


def rccl_shard(world, rank, shard_boards=None, total_boards=None, group=None, lib=None):
    """The `shard=` argument of NativeSolver for the library's OWN exchange (prl_solver_create_sharded_rccl): rank 0 draws the
    communicator id (ncclGetUniqueId), torch.distributed broadcasts its 128 bytes, every rank joins inside the library; from then on the
    all-gather of every EV pass is one ncclAllGather on the solver's stream -- no Python, no callback in the iteration loop.
    torch.distributed is used for this one broadcast only (any other channel would do: the C ABI takes the raw bytes)."""
    import ctypes
    import os
    from pokerrl_amd import _native
    L = lib or _native.lib()
    if world > 1:
        # PRE-FLIGHT, before anybody enters a collective of the new communicator: every rank says whether IT can bind RCCL (prl_rccl_info) and the
        # ranks agree (all-reduce MIN over the torch group). A rank that cannot (a mistyped PRL_RCCL_LIB on one host, a second ROCm) makes ALL
        # ranks raise here -- otherwise the others would sit in ncclCommInitRank waiting for it. (PRL_TEST_RCCL_FAIL_RANK=<r>: the tests' way
        # of failing one rank.)
        info = ctypes.create_string_buffer(512)
        ok = int(L.prl_rccl_info(info, 512)) == 0
        why = info.value.decode("utf-8", "replace")
        if os.environ.get("PRL_TEST_RCCL_FAIL_RANK") == str(rank):
            ok, why = False, "forced by PRL_TEST_RCCL_FAIL_RANK"
        dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        if int(flag.item()) == 0:
            raise RuntimeError("rccl_shard: RCCL cannot be bound on every rank (rank %d: %s); every rank raises before any ncclCommInitRank -- "
                               "fall back to exchange='torch'" % (rank, why if not ok else "fine here: " + why))
    buf = (ctypes.c_char * 128)()
    # every rank enters the broadcast whatever happened on rank 0: a status byte travels with the id, so that a rank 0 that could not
    # draw one (librccl not loadable) makes ALL ranks raise instead of leaving the others inside the collective
    status, err = 0, ""
    if rank == 0:
        status = int(L.prl_rccl_unique_id(ctypes.cast(buf, ctypes.c_void_p)))
        if status != 0:
            err = L.prl_last_error().decode("utf-8", "replace")
    if world > 1:
        dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
        t = torch.frombuffer(bytearray(bytes(buf) + bytes([1 if status != 0 else 0])), dtype=torch.uint8).clone().to(dev)
        dist.broadcast(t, src=0, group=group)
        raw = bytes(t.cpu().numpy().tobytes())
        uid, failed = raw[:128], raw[128] != 0
    else:
        uid, failed = bytes(buf), status != 0
    if failed:
        raise RuntimeError("rccl_shard: rank 0 could not draw an RCCL communicator id" + (": " + err if err else "") +
                           " (every rank raises; fall back to exchange='torch')")
    if shard_boards is None:
        return ("rccl", world, rank, uid)
    return ("rccl", world, rank, uid, int(shard_boards), int(total_boards))
