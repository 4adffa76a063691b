"""
Pre-flight of an N-GPU node BEFORE the scaling bench (no 8-GPU node was available to the builder: RCCL with more than one rank has never
executed this code). One process per GPU, launched like the bench:

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29517 scripts/scale_preflight.py [--gb 88]

Every step prints `[preflight] rank r: <step>: ok | FAILED: why` and the script ends with ONE json line on rank 0; any failure -> exit code 1
with the step named, so that a hang or a crash inside the full bench is never the first sign of trouble. Steps:
  1 rccl_info        which librccl.so the library binds in this process (prl_rccl_info), the same path on every rank
  2 comm_init        rccl_shard's agreement + ncclCommInitRank over all ranks through a tiny sharded solver (32 boards per rank, one iteration)
  3 all_gather       the REAL exchange size of bench.py's default run (256 groups x 3 root vectors x 1326 floats = 4.07 MB per rank) through the same
                     solver path: a 262144-board-per-rank geometry is not needed for that -- the exchange buffer of `--boards 32768` is the same shape
  4 vmm_alloc        --gb gigabytes (default 88 = --all-boards per rank) of shuffled 2 MB virtual-memory backing (PRL_VMM_SHUFFLE_MB=2), touched, freed
  5 bench_smoke      bench.py's own code path at 32768 boards per rank, 3 steps (about 30 s), exploitability and the fixed-problem bits printed
"""
import argparse
import ctypes
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gb", type=float, default=88.0)
    ap.add_argument("--skip-bench", action="store_true")
    args = ap.parse_args()
    rank, local, world = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(local)
    dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    from pokerrl_amd import _native
    from pokerrl_amd.dist import rccl_shard
    import bench
    _native.require_device()
    _native.set_device(local)
    L = _native.lib()
    report, failed = {"world": world}, []

    def step(name, fn):
        t0 = time.perf_counter()
        try:
            val = fn()
            ok, why = True, ""
        except Exception as e:  # noqa: BLE001  (every failure is reported by name, on every rank)
            val, ok, why = None, False, "%s: %s" % (type(e).__name__, e)
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        all_ok = int(flag.item()) == 1
        print("[preflight] rank %d: %s: %s (%.1f s)" % (rank, name, "ok" if ok else "FAILED: " + why, time.perf_counter() - t0), flush=True)
        report[name] = {"ok": all_ok, "value": val if ok else why}
        if not all_ok:
            failed.append(name)
        return val if all_ok else None

    def rccl_info():
        buf = ctypes.create_string_buffer(512)
        rc = L.prl_rccl_info(buf, 512)
        if rc != 0:
            raise RuntimeError(buf.value.decode("utf-8", "replace"))
        paths = [None] * world
        dist.all_gather_object(paths, buf.value.decode())
        if len(set(paths)) != 1:
            raise RuntimeError("the ranks bind different RCCL libraries: %s" % paths)
        return paths[0]

    step("rccl_info", rccl_info)

    def comm_init():
        tree = bench.fhp_tree(bench.seeded_boards(32, 0, offset=rank * 32))
        s = _native.NativeSolver(tree, "plus", 0, shard=rccl_shard(world, rank))
        s.iterations(1)
        e = np.asarray(s.exploitability(), np.float32)
        bits = [None] * world
        dist.all_gather_object(bits, e.tobytes().hex())
        if len(set(bits)) != 1:
            raise RuntimeError("the ranks disagree on the exploitability: %s" % bits)
        return bits[0]

    if not failed:
        step("comm_init", comm_init)

    def all_gather():
        tree = bench.fhp_tree(bench.seeded_boards(32768, 0, offset=rank * 32768))
        s = _native.NativeSolver(tree, "plus", 0, shard=rccl_shard(world, rank))
        s.sync()
        t0 = time.perf_counter()
        s.iterations(2)
        s.sync()
        return {"exchanges": int(s.get("exchanges")[0]), "ms_per_iteration": (time.perf_counter() - t0) * 500.0,
                "exploitability_mbb_per_g": float(np.mean(s.exploitability()) * 10.0)}

    if not failed:
        step("all_gather", all_gather)

    def vmm_alloc():
        # a child process per rank: PRL_VMM_SHUFFLE_MB is read at library start, and an allocation failure must not take this process down
        code = ("import sys; sys.path.insert(0, %r)\nimport torch\ntorch.cuda.set_device(%d)\nfrom pokerrl_amd import _native\n_native.set_device(%d)\n"
                "import bench\nn = int(%f * 1e9 / 221e3)\nt = bench.fhp_tree(bench.seeded_boards(n, 0))\ns = _native.NativeSolver(t, 'plus', 0, engine='fused')\n"
                "s.iterations(1); s.sync()\nprint(int(s.get('bytes_allocated')[0]))\n" % (ROOT, local, local, args.gb))
        env = dict(os.environ, PRL_VMM_SHUFFLE_MB="2")
        for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=900)
        if out.returncode != 0:
            raise RuntimeError(out.stderr[-400:])
        return {"bytes_allocated": int(out.stdout.strip().splitlines()[-1])}

    if not failed and args.gb > 0:
        step("vmm_alloc", vmm_alloc)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0 and not failed and not args.skip_bench:
        # the bench's own launcher, the way the driver starts it for N = 1 (it starts its ranks itself)
        t0 = time.perf_counter()
        env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "TORCHELASTIC_RUN_ID")}
        out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(world), "--boards", "32768", "--steps", "3", "--warmup", "1",
                              "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
        line = [x for x in out.stdout.splitlines() if x.startswith("{")]
        ok = out.returncode == 0 and len(line) == 1
        j = json.loads(line[0]) if ok else None
        report["bench_smoke"] = {"ok": ok, "value": {"value": j["value"], "ms_per_step": j["ms_per_step"], "exchange": j["config"]["exchange"],
                                                     "rccl_library": j["config"]["rccl_library"], "fixed_problem_check": j["config"]["fixed_problem_check"],
                                                     "per_rank_ms_per_step": j["config"]["per_rank_ms_per_step"]} if ok else out.stderr[-600:]}
        print("[preflight] bench_smoke: %s (%.1f s)" % ("ok" if ok else "FAILED", time.perf_counter() - t0), flush=True)
        if not ok:
            failed.append("bench_smoke")
    if rank == 0:
        report["failed"] = failed
        print(json.dumps(report), flush=True)
    sys.exit(1 if failed else 0)


if __name__ == "__main__":
    main()
