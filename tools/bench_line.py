"""Condense a bench.py JSON line (stdin) to: ms/step, value, roofline.frac, step fraction of the MFMA peak, final loss."""
import json
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else ""
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = j.get("roofline", {})
print(tag, j["ms_per_step"], "ms", j["value"], j["unit"], "frac", r.get("frac"), "step_frac", r.get("step_frac_of_mfma_peak"), "final_loss", j["config"].get("final_loss"))
