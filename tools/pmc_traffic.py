"""HBM traffic of the GEMM launches from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; csv output), corrected as
/opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950: FETCH_SIZE is reported in KiB and counts 128-B requests at
64 B for wide coalesced reads -> doubled; WRITE_SIZE is taken as reported (uncalibrated, stated as such)."""
import csv, sys, collections

def load(path):
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            if "gemm256" in r["Kernel_Name"]:
                rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"].split("(")[0].split("::")[-1], int(r["Grid_Size"]), float(r["Counter_Value"])))
    return rows

fetch, write = load(sys.argv[1]), load(sys.argv[2])
# the micro-benchmark launches every shape 12 times back to back: group consecutive dispatches of the same kernel/grid
def groups(rows):
    out, cur = [], None
    for d, k, g, v in rows:
        if cur and cur[0] == (k, g):
            cur[1].append(v)
        else:
            cur = [(k, g), [v]]
            out.append(cur)
    return out
names = sys.argv[3].split(",") if len(sys.argv) > 3 else None
gf, gw = groups(fetch), groups(write)
print("| GEMM | launches | FETCH_SIZE raw (MB) | read traffic = 2 x FETCH (MB) | WRITE_SIZE (MB) |")
print("|---|---|---|---|---|")
for i, (a, b) in enumerate(zip(gf, gw)):
    f = sum(a[1][2:]) / max(1, len(a[1][2:])) / 1024.0   # skip the two warm-up launches; KiB -> MiB ~ MB
    w = sum(b[1][2:]) / max(1, len(b[1][2:])) / 1024.0
    nm = names[i] if names and i < len(names) else f"{a[0][0]} grid {a[0][1]}"
    print(f"| {nm} | {len(a[1])} | {f:.0f} | {2 * f:.0f} | {w:.0f} |")
