"""profiles/lbr_counters.json from a rocprofv3 kernel-trace + PMC run of bench_lbr.py (scripts/gpu_r6_lbr.sh): the SQ counters and the mean launch
duration of prl_k_lbr_batch, per launch -- what bench_lbr.py's roofline object is computed from (no literal in the bench).

    python scripts/lbr_counters.py <stats db> <counter_collection.csv>... --hands H --cmd "..." > gpurun_out/TAG_lbr_counters.json
"""
import csv
import json
import sqlite3
import sys
from collections import defaultdict


def main(argv):
    hands, cmd, tag, paths = None, "", "", []
    it = iter(argv)
    for a in it:
        if a == "--hands":
            hands = int(next(it))
        elif a == "--cmd":
            cmd = next(it)
        elif a == "--tag":
            tag = next(it)
        else:
            paths.append(a)
    out = {"kernel": None, "hands_per_launch": hands, "command": cmd, "checkpoint": tag, "counters": {}, "launches": {}}
    for p in paths:
        if p.endswith(".db"):
            c = sqlite3.connect(p)
            for name, calls, total, avg in c.execute("select name, total_calls, total_duration, average from top_kernels"):
                if "prl_k_lbr_batch" in name:
                    out["kernel"] = name
                    out["kernel_us_mean"] = float(avg)  # microseconds (scripts/rocprof_summary.py prints the same column)
                    out["kernel_calls"] = int(calls)
            continue
        agg = defaultdict(list)
        with open(p) as f:
            for row in csv.DictReader(f):
                if "prl_k_lbr_batch" in row["Kernel_Name"]:
                    agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, v in agg.items():
            out["counters"][k] = sum(v) / len(v)  # bench_lbr.py --no-warmup launches the kernel twice (the two seats), both of hands_per_launch hands
            out["launches"][k] = len(v)
    json.dump(out, sys.stdout, indent=1)
    sys.stdout.write("\n")


if __name__ == "__main__":
    main(sys.argv[1:])
