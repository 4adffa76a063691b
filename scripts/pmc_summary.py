"""Aggregate rocprofv3 --pmc counter_collection.csv files: mean counter value per dispatch, per kernel name."""
import csv
import sys
from collections import defaultdict


def main(paths):
    for p in paths:
        agg = defaultdict(lambda: defaultdict(list))
        with open(p) as f:
            for row in csv.DictReader(f):
                agg[row["Kernel_Name"][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        print("==", p)
        for k, cs in agg.items():
            n = max(len(v) for v in cs.values())
            print("%-62s n=%d " % (k, n) + "  ".join("%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(cs.items())))


if __name__ == "__main__":
    main(sys.argv[1:])
