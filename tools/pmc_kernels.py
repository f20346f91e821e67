"""Per-kernel averages of the counters in rocprofv3 --pmc csv outputs, for kernels whose name contains a pattern.

  python tools/pmc_kernels.py attn_bwd <counter_collection.csv> [more.csv ...]
"""
import collections
import csv
import sys

pat = sys.argv[1]
for path in sys.argv[2:]:
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(dict)
    with open(path) as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"]
            if pat not in k:
                continue
            k = k.replace("(anonymous namespace)::", "").split("(")[0][-60:]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[k][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    for k, cs in acc.items():
        d = sorted(dur[k].values())
        print(f"{k}  launches {len(d)}  median {d[len(d) // 2]:.1f} us")
        for c, v in sorted(cs.items()):
            print(f"    {c:32s} {sum(v) / len(v):16.0f}")
