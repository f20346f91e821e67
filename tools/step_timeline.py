"""Where one multi-stream step's wall time goes, from a rocprofv3 --kernel-trace rocpd database of `bench.py` (shipped five-stream schedule).

Steps are cut at the optimizer kernel (`adamw_kernel`, one dispatch per step); for the last complete step the tool prints
  * wall time, the sum of kernel durations, the union of the kernel intervals (GPU busy), time with 1 / 2 / >= 3 kernels in flight,
  * per kernel family: time during which it was the ONLY kernel in flight (what the step pays in full) against its total duration,
  * the largest idle gaps with the kernels either side,
  * per hardware queue: dispatches and busy time.

    python tools/step_timeline.py <db> [top]
"""
import re
import sqlite3
import sys
from collections import defaultdict


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"void ", "", name)
    name = re.sub(r"\(.*", "", name)
    return name[:90]


def main(db: str, top: int = 24) -> None:
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    qcol = next((q for q in ("queue_id", "queue", "stream_id", "stream") if q in cols), None)
    sel = "name, start, end" + (f", {qcol}" if qcol else "")
    rows = sorted(c.execute(f"select {sel} from kernels").fetchall(), key=lambda r: r[1])
    marks = [r[2] for r in rows if "adamw_kernel" in r[0]]
    if len(marks) < 2:
        print("fewer than two optimizer dispatches in the trace")
        return
    # the EMA follows AdamW: a step = (end of the previous step's last optimizer-phase kernel, end of this step's AdamW]
    t0, t1 = marks[-2], marks[-1]
    step = [r for r in rows if r[2] > t0 and r[1] < t1]
    ev = []
    for i, r in enumerate(step):
        ev.append((max(r[1], t0), 1, i))
        ev.append((min(r[2], t1), -1, i))
    ev.sort()
    live: set = set()
    last = t0
    by_depth = defaultdict(float)
    alone = defaultdict(float)
    total = defaultdict(float)
    gaps = []
    prev_name = "(step start)"
    for t, d, i in ev:
        dt = t - last
        if dt > 0:
            n = len(live)
            by_depth[min(n, 3)] += dt
            if n == 1:
                alone[short(step[next(iter(live))][0])] += dt
            if n == 0:
                gaps.append((dt, prev_name, None, last - t0))
        last = t
        if d == 1:
            if not live and gaps and gaps[-1][2] is None:
                gaps[-1] = (gaps[-1][0], gaps[-1][1], short(step[i][0]), gaps[-1][3])
            live.add(i)
        else:
            live.discard(i)
            prev_name = short(step[i][0])
    for r in step:
        total[short(r[0])] += min(r[2], t1) - max(r[1], t0)
    wall = t1 - t0
    ms = 1e-6
    print(f"columns of `kernels`: {cols}\n")
    print(f"last step: wall {wall*ms:.2f} ms, {len(step)} dispatches, sum of kernel durations {sum(total.values())*ms:.2f} ms")
    print(f"  idle (no kernel in flight) {by_depth[0]*ms:.2f} ms | exactly 1 kernel {by_depth[1]*ms:.2f} | 2 kernels {by_depth[2]*ms:.2f} | >= 3 kernels {by_depth[3]*ms:.2f}\n")
    print("| kernel | total ms | ms as the only kernel in flight |")
    print("|---|---|---|")
    for k, t in sorted(total.items(), key=lambda kv: -kv[1])[:top]:
        print(f"| `{k}` | {t*ms:.2f} | {alone.get(k, 0.0)*ms:.2f} |")
    print("\nlargest idle gaps (us, at ms into the step, after -> before):")
    for dt, a, b, at in sorted(gaps, reverse=True)[:12]:
        print(f"  {dt/1e3:8.1f} us at {at*ms:6.2f} ms: {a} -> {b}")
    if len(sys.argv) > 3:   # python tools/step_timeline.py <db> <top> <kernel-substring> [ms before] [ms after]: the launches around the first match
        pat = sys.argv[3]
        before = float(sys.argv[4]) if len(sys.argv) > 4 else 3.0
        after = float(sys.argv[5]) if len(sys.argv) > 5 else 4.0
        hit = next((r for r in step if pat in r[0]), None)
        if hit is not None:
            print(f"\nlaunches from {before} ms before to {after} ms after the start of `{pat}` (start offset ms | duration us | {qcol} | kernel):")
            for r in step:
                if hit[1] - before * 1e6 <= r[1] <= hit[1] + after * 1e6:
                    print(f"  {(r[1] - hit[1]) * ms:8.3f} | {(r[2] - r[1]) / 1e3:8.1f} | {r[3] if qcol else '-'} | {short(r[0])}")
    if qcol:
        q = defaultdict(lambda: [0, 0.0])
        for r in step:
            q[r[3]][0] += 1
            q[r[3]][1] += r[2] - r[1]
        print(f"\nper {qcol}: " + ", ".join(f"{k}: {n} dispatches {t*ms:.1f} ms" for k, (n, t) in sorted(q.items(), key=lambda kv: str(kv[0]))))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 24)
