"""Kernels of one step around a marker kernel, with start / end times (ms into the step) and hardware queue, from a rocprofv3 --kernel-trace database:
what runs on which stream while the step waits for that kernel.   python tools/step_window.py <db> <marker substring> [ms before] [ms after]"""
import re, sqlite3, sys

def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"void ", "", n); n = re.sub(r"\(.*", "", n)
    return n[:70]

db, marker = sys.argv[1], sys.argv[2]
before = float(sys.argv[3]) if len(sys.argv) > 3 else 3.0
after = float(sys.argv[4]) if len(sys.argv) > 4 else 2.0
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
qcol = next((q for q in ("queue_id", "queue", "stream_id", "stream") if q in cols), None)
rows = sorted(c.execute(f"select name, start, end, {qcol} from kernels").fetchall(), key=lambda r: r[1])
marks = [r[2] for r in rows if "adamw_kernel" in r[0]]
t0, t1 = marks[-2], marks[-1]
step = [r for r in rows if r[2] > t0 and r[1] < t1]
m = [r for r in step if marker in r[0]]
if not m:
    sys.exit(f"no kernel matching {marker!r} in the last step")
tm = m[0][1]
print(f"step of {(t1 - t0) / 1e6:.2f} ms; first `{marker}` starts {(tm - t0) / 1e6:.2f} ms into it")
print("| start ms | end ms | us | queue | kernel |")
print("|---|---|---|---|---|")
for n, s, e, q in step:
    if e > tm - before * 1e6 and s < tm + after * 1e6:
        print(f"| {(s - t0) / 1e6:7.3f} | {(e - t0) / 1e6:7.3f} | {(e - s) / 1e3:7.1f} | {q} | `{short(n)}` |")
