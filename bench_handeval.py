"""
bench_handeval.py -- 7-card hand evaluations/s: all 1326 hole-card pairs on B seeded 5-card boards, device buffers in and out
(prl_hand_rank_boards_device = get_hand_rank_all_hands_on_given_boards_52_holdem of the reference, CppHandeval.py:34-46).
Algorithmic bytes: 5 B of board in + 4 B per (board, hand) out; the kernel is bound by integer ALU issue and the int32 store
(SURVEY.md section 8d). The CPU figure beside it is the C oracle's evaluator on one core (the same run), with the REFERENCE BINARY's batched rate on
one core of the build container next to it (profiles/reference_cpu.json: lib_hand_eval.so does not travel either) -- the faster of the two baselines.

    python bench_handeval.py [--boards B] [--reps K]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import bench_ref  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--boards", type=int, default=1 << 17)
    ap.add_argument("--reps", type=int, default=10)
    args = ap.parse_args()
    import torch
    import bench
    import oracle
    from pokerrl_amd import _native
    _native.require_device()
    L = _native.lib()
    boards = bench.seeded_boards(args.boards, 0)
    d_b = torch.from_numpy(boards.astype(np.int8)).cuda()
    d_o = torch.empty((args.boards, 1326), dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    import ctypes
    L.prl_hand_rank_boards_device.argtypes = [ctypes.c_void_p, ctypes.c_int32, ctypes.c_void_p, ctypes.c_void_p]

    def run():
        assert L.prl_hand_rank_boards_device(d_b.data_ptr(), args.boards, d_o.data_ptr(), None) == 0

    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.reps
    evals = args.boards * 1326
    n_cpu = 2000
    t0 = time.perf_counter()
    ref = oracle.rank_boards(boards[:n_cpu])
    cpu = n_cpu * 1326 / (time.perf_counter() - t0)
    assert np.array_equal(d_o[:n_cpu].cpu().numpy(), ref)
    bytes_call = args.boards * (5.0 + 4.0 * 1326)  # algorithmic: the board in, one int32 rank per (board, hand) out
    achieved = bytes_call / (ms * 1e-3) / 1e9
    print(json.dumps({"metric": "7-card hand evaluations/s (all 1326 hands on given boards)", "value": evals / (ms * 1e-3), "unit": "evals/s",
                      "n_gpus": 1, "steps": args.reps, "warmup": 1, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                      "dtype": "int32", "data": "synthetic", "build_flavor": _native.build_flavor(),
                      "config": {"workload": "ranks of all 1326 hole-card pairs on %d seeded 5-card boards per call, device buffers in and out" % args.boards,
                                 "boards": args.boards},
                      "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s", "frac": achieved / 8000.0, "traffic": None,
                                   "kernel": "prl_k_hand_rank_boards", "kernel_ms_per_launch": ms, "bytes_per_call_algorithmic": bytes_call,
                                   # SQ counters of this kernel (profiles/r50_handeval_pmc.txt, r50_handeval_kernel_stats.txt): 3.892e8 VALU + 1.644e8 SALU
                                   # wave-instructions per launch of 131072 boards in 0.581 ms; a wave64 VALU instruction issues over 2 cycles on a 32-lane SIMD
                                   "valu_issue_busy_measured": 3.892e8 * 2.0 / (1024 * 0.581e-3 * 2.4e9),
                                   "pmc_traffic_bytes_per_call": 6.82e8 + 2.0 * 2.634e6,
                                   "note": "bound by integer vector issue: the VALU issue slots are 0.55 busy (measured) where the int32 result stream is 0.15 of HBM; "
                                           "HBM traffic = the algorithmic bytes (WRITE_SIZE 682 MB per 695 MB call)"},
                      "cpu_baseline": {"value": cpu, "unit": "evals/s", "cores": 1, "kind": "port", "sample": "%d boards, oracle/prl_oracle.c" % n_cpu,
                                       # the reference's own evaluator (binary-only lib_hand_eval.so, batched call): timed by scripts/time_reference.py
                                       "reference_binary_evals_per_s_per_core": bench_ref.figure("hand_evaluator", "batched", "evals_per_s"),
                                       "reference_timing_source": bench_ref.SOURCE, "reference_timing_host": bench_ref.host()}}))


if __name__ == "__main__":
    main()
