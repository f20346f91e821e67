"""Summarise a rocprofv3 --kernel-trace rocpd sqlite database into a per-kernel table (markdown)."""
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"void ", "", name)
    return name[:110]


def main(db: str, top: int = 40) -> None:
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    rows = c.execute("select name, start, end from kernels").fetchall()
    agg = {}
    for name, s, e in rows:
        d = agg.setdefault(short(name), [0, 0.0])
        d[0] += 1
        d[1] += (e - s)
    tot = sum(v[1] for v in agg.values())
    print(f"| kernel | calls | total ms | avg us | % |")
    print("|---|---|---|---|---|")
    for k, (n, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"| `{k}` | {n} | {t/1e6:.2f} | {t/n/1e3:.1f} | {100*t/tot:.1f} |")
    print(f"\ntotal kernel time {tot/1e6:.1f} ms over {len(rows)} dispatches")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
