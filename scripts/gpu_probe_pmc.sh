#!/bin/bash
# per-dispatch counters of the board pass over several solver objects of one process (scripts/gpu_mode_probe.py): which counter follows the
# fast / slow placement? usage: scripts/gpu_probe_pmc.sh tag "COUNTER ..." ["COUNTER ..."]
cd $GRAFT_REPO_ROOT; TAG=$1; shift; R=$GRAFT_REPO_ROOT; mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "$@"; do
  i=$((i+1))
  PROBE_REPS=1 timeout 600 rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/${TAG}_p$i -o p --output-format csv -- python $R/scripts/gpu_mode_probe.py 5 > $R/gpurun_out/${TAG}_p$i.log 2>&1
  grep solver $R/gpurun_out/${TAG}_p$i.log
  python - <<PY
import csv, glob, collections
f = glob.glob("$R/gpurun_out/${TAG}_p$i/**/*counter_collection.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "fhp_pass" in r["Kernel_Name"] and ("ILi5ELi0ELi0ELb1" in r["Kernel_Name"] or "<5, 0, 0, true>" in r["Kernel_Name"])]
by = collections.OrderedDict()
for r in rows:
    by.setdefault(r["Dispatch_Id"], {})[r["Counter_Name"]] = float(r["Counter_Value"])
    by[r["Dispatch_Id"]]["_t"] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e6 if "End_Timestamp" in r else 0.0
ids = list(by)
print("dispatches of pass<5,steady>:", len(ids))
names = sorted(k for k in by[ids[0]] if k != "_t")
print("disp  ms    " + "  ".join(names))
for d in ids[::3]:
    print(d, "%.2f" % by[d]["_t"], "  ".join("%.4g" % by[d].get(n, -1) for n in names))
PY
  rm -rf $R/gpurun_out/${TAG}_p$i
done
