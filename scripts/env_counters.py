"""profiles/env_counters.json from rocprofv3 --pmc runs of bench_env.py (one counter group per run, --kernel-trace only): mean per dispatch of
prl_k_ebf_random_step at 2^20 envs. bench_env.py reads its roofline.traffic from this file (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes
for gfx950, WRITE_SIZE as printed; both in KB).   python scripts/env_counters.py <tag> <out.json> <counter_collection.csv> ..."""
import csv
import json
import sys
from collections import defaultdict


def main(tag, out, paths):
    agg = defaultdict(list)
    for p in paths:
        with open(p) as f:
            for row in csv.DictReader(f):
                if "prl_k_ebf_random_step" in row["Kernel_Name"]:
                    agg[row["Counter_Name"]].append(float(row["Counter_Value"]))
    c = {k: sum(v) / len(v) for k, v in agg.items()}
    d = {"source": "rocprofv3 --kernel-trace --pmc <one group per run> -- python bench_env.py --steps 40 --warmup 5 --no-cpu-baseline (2^20 envs), checkpoint " + tag,
         "kernel": "prl_k_ebf_random_step", "envs": 1 << 20, "dispatches": {k: len(v) for k, v in agg.items()}, "counters_mean_per_dispatch": c}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        d["hbm_read_bytes"] = 2.0 * c["FETCH_SIZE"] * 1024.0
        d["hbm_write_bytes"] = c["WRITE_SIZE"] * 1024.0
        d["hbm_bytes_per_env_step"] = (d["hbm_read_bytes"] + d["hbm_write_bytes"]) / (1 << 20)
    if "SQ_INSTS_VALU" in c and "SQ_WAVES" in c:
        d["valu_instructions_per_wave"] = c["SQ_INSTS_VALU"] / c["SQ_WAVES"]
        d["salu_instructions_per_wave"] = c.get("SQ_INSTS_SALU", 0.0) / c["SQ_WAVES"]
    if "SQ_WAIT_ANY" in c and "SQ_WAVE_CYCLES" in c:
        d["wait_any_over_wave_cycles"] = c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]
    json.dump(d, open(out, "w"), indent=1)
    print(json.dumps(d))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], sys.argv[3:])
