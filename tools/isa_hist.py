"""Instruction histogram of one kernel in a hipcc -save-temps .s file (diagnostic)."""
import re, sys
from collections import Counter
path, pat = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = [i for i, l in enumerate(lines) if re.match(r"^_Z\S*" + pat + r"\S*: ", l)][0]
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
c = Counter()
loop = None
for l in lines[start:end]:
    t = l.strip()
    if not t or t.startswith((".", ";")) or t.endswith(":") or t.startswith("_Z"):
        continue
    c[t.split()[0]] += 1
for k, v in c.most_common(int(sys.argv[3]) if len(sys.argv) > 3 else 40):
    print(f"{k:34s}{v}")
print("total", sum(c.values()), "lines", end - start)
