"""profiles/multistreet_counters.json from rocprofv3 PMC runs of bench_multistreet.py (scripts/gpu_r6_ms_traffic.sh): HBM bytes of the last street's
steady-state passes per CFR+ iteration, per workload -- what bench_multistreet.py's roofline.traffic is read from (no literal in the bench).

    python scripts/multistreet_counters.py KEY <bench json line file> <FETCH_SIZE counter_collection.csv> <WRITE_SIZE counter_collection.csv> --cmd "..." --tag TAG [--into existing.json]

FETCH_SIZE and WRITE_SIZE come from separate runs (--kernel-trace only beside --pmc), KB as rocprofv3 prints them; on gfx950 FETCH_SIZE counts a 128-byte
request as 64 bytes: doubled here (MI355X_MICROARCH.md, HBM section). The steady passes are prl_k_st_pass<true, 4 | 5, ...> (one launch per last-street
(street, shape) group, seat and iteration); the mean per dispatch of every such kernel name, summed over the names = one iteration.
"""
import csv
import json
import re
import sys
from collections import defaultdict


def means(path, counter):
    agg = defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == counter and re.search(r"prl_k_st_pass<true, [45],", row["Kernel_Name"]):
                agg[row["Kernel_Name"].split("(")[0]].append(float(row["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in agg.items()}


def main(argv):
    key, bench_file, fetch_csv, write_csv = argv[:4]
    cmd = tag = into = ""
    it = iter(argv[4:])
    for a in it:
        if a == "--cmd":
            cmd = next(it)
        elif a == "--tag":
            tag = next(it)
        elif a == "--into":
            into = next(it)
    line = [ln for ln in open(bench_file).read().splitlines() if ln.startswith("{")][-1]
    bench = json.loads(line)
    fetch, write = means(fetch_csv, "FETCH_SIZE"), means(write_csv, "WRITE_SIZE")
    kernels, total = {}, 0.0
    for name in sorted(set(fetch) | set(write)):
        f_kb, f_n = fetch.get(name, (0.0, 0))
        w_kb, w_n = write.get(name, (0.0, 0))
        kernels[name] = {"fetch_size_kb_mean": f_kb, "fetch_dispatches": f_n, "write_size_kb_mean": w_kb, "write_dispatches": w_n}
        total += (2.0 * f_kb + w_kb) * 1024.0
    cols = bench["config"]["action_columns_last_street"]
    entry = {"workload": bench["config"]["workload"], "command": cmd, "checkpoint": tag, "kernels": kernels,
             "hbm_bytes_per_iteration_last_street": total, "action_columns_last_street": cols, "hbm_bytes_per_last_street_column": total / cols,
             "algorithmic_bytes_per_iteration_last_street": bench["roofline"]["bytes_per_iteration_algorithmic"],
             "traffic_over_algorithmic": total / bench["roofline"]["bytes_per_iteration_algorithmic"]}
    out = {}
    if into:
        try:
            out = json.load(open(into))
        except (OSError, ValueError):
            out = {}
    out[key] = entry
    json.dump(out, sys.stdout, indent=1)
    sys.stdout.write("\n")


if __name__ == "__main__":
    main(sys.argv[1:])
