"""Per-launch HBM-side traffic of the GEMM kernels over one bench step, from two rocprofv3 --pmc passes
(FETCH_SIZE, WRITE_SIZE) of `bench.py --steps 1 --warmup 1 --single-stream`.  gfx950 correction per
/opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE (KiB) is doubled (128-B requests tallied at 64 B for wide
coalesced reads); WRITE_SIZE (KiB) is used as reported (it matched the algorithmic store bytes of six isolated
GEMM shapes within 1 %, profiles/r01h_gemm_traffic.md)."""
import csv, json, os, re, sys, collections

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def load(path):
    per = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            n = r["Kernel_Name"]
            m = re.search(r"(gemm\w*_kernel<[^>]*>)", n)
            if m:
                per[m.group(1)].append(float(r["Counter_Value"]) * 1024.0)
    return per

fetch, write = load(sys.argv[1]), load(sys.argv[2])
steps = 2   # 1 warm-up + 1 timed step in the profiled command
tot_f = sum(sum(v) for v in fetch.values()) * 2.0
tot_w = sum(sum(v) for v in write.values())
n = sum(len(v) for v in fetch.values())
print("| kernel | launches/step | read MB/launch (2 x FETCH_SIZE) | write MB/launch |")
print("|---|---|---|---|")
for k in sorted(fetch, key=lambda k: -sum(fetch[k])):
    print(f"| `{k}` | {len(fetch[k]) // steps} | {2 * sum(fetch[k]) / len(fetch[k]) / 1e6:.0f} | {sum(write[k]) / len(write[k]) / 1e6:.0f} |")
import bench  # noqa: E402  (kernel_sha16: the stamp bench.py checks before quoting this measurement)

out = {"kernel_sha16": bench.kernel_sha16(), "profile": sys.argv[3] if len(sys.argv) > 3 else "profiles/gemm_traffic.json",
       "gemm_launches_per_step": n // steps, "read_bytes_per_launch": tot_f / n, "write_bytes_per_launch": tot_w / n,
       "traffic_bytes_per_launch": (tot_f + tot_w) / n, "traffic_bytes_per_step": (tot_f + tot_w) / steps,
       "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), bench.py --steps 1 --warmup 1 --single-stream; FETCH_SIZE doubled (gfx950)"}
print()
print(json.dumps(out))
