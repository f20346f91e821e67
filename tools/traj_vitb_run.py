import sys, os, json
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, os.path.join(ROOT, "tests", "tools")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import trajectory
for koleo in (0.0, 0.1):
    worst, rows, own = trajectory.run_vs_reference(koleo, 100, quiet=True, fixture="vitb")
    print(koleo, json.dumps(worst), "first/last loss", rows[0][1], rows[-1][1], "ref", rows[0][2], rows[-1][2])
    print("  worst step:", max(rows, key=lambda r: r[3]["loss"])[0], [f"{r[3]['loss']:.1e}" for r in rows[::10]])
