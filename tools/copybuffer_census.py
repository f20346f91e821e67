"""Where do the __amd_rocclr_copyBuffer dispatches of a step come from?  For every such dispatch of a rocprofv3 --kernel-trace database: the
kernels dispatched just before and just after it on the same stream, and its grid size; the (before -> after) pairs are counted.

  python tools/copybuffer_census.py <results.db> [single-stream run: the neighbours are then exact]"""
import collections
import re
import sqlite3
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"void ", "", name)
    return name.split("(")[0][-70:]


c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end, stream_id, grid_x, workgroup_x from kernels order by start").fetchall()
by_stream = collections.defaultdict(list)
for r in rows:
    by_stream[0].append(r)      # (the runtime's copy kernels carry a stream id of their own: neighbours are taken in time order over all streams)
pairs = collections.Counter()
grids = collections.Counter()
n = 0
for st, lst in by_stream.items():
    for i, r in enumerate(lst):
        if "copyBuffer" not in r[0]:
            continue
        n += 1
        j = i - 1
        while j >= 0 and "copyBuffer" in lst[j][0]:
            j -= 1
        k = i + 1
        while k < len(lst) and "copyBuffer" in lst[k][0]:
            k += 1
        prev = short(lst[j][0]) if j >= 0 else "-"
        nxt = short(lst[k][0]) if k < len(lst) else "-"
        pairs[(prev, nxt)] += 1
        grids[(r[4], r[5])] += 1
print(f"{n} copyBuffer dispatches on {len(by_stream)} streams; grid sizes: {dict(grids.most_common(6))}")
for (a, b), k in pairs.most_common(25):
    print(f"{k:5d}  after `{a}`  before `{b}`")
