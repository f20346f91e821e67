"""Copy the outputs of `tools/gpu_profiles.sh <tag>` (gpurun_out/<tag>/, scratch) into profiles/ under the round's file names."""
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
src = os.path.join(ROOT, "gpurun_out", tag)
names = {"bench_default_full.log": f"{tag}_bench_default.json.log", "kernel_stats_single.md": f"{tag}_kernel_stats_single.md",
         "kernel_stats_multi.md": f"{tag}_kernel_stats_multi.md", "pmc_step_report.md": f"{tag}_pmc_step_report.md", "gemm_traffic.txt": f"{tag}_gemm_traffic.txt",
         "gemm_timeline.txt": f"{tag}_gemm_timeline.txt", "bench_vits.log": f"{tag}_bench_vit_small.json.log", "cfg5.log": f"{tag}_cfg5_run.log",
         "bench_cfg4_resnet50.log": f"{tag}_bench_distillationv3_resnet50.json.log", "bench_gloo2.log": f"{tag}_bench_gloo2_one_gpu.json.log",
         "gpu_tests_tail.log": f"{tag}_gpu_tests_tail.log", "step_timeline.txt": f"{tag}_step_timeline.txt", "kernel_stats_vits.md": f"{tag}_kernel_stats_vits.md",
         "bench_default_b.log": f"{tag}_bench_default_b.json.log", "bench_default_c.log": f"{tag}_bench_default_c.json.log"}
for a, b in names.items():
    p = os.path.join(src, a)
    if os.path.exists(p):
        lines = [ln for ln in open(p, errors="replace").read().splitlines() if "amdgpu.ids" not in ln]
        if a.startswith("bench_") or a == "cfg5.log":
            lines = lines[-4:]
        open(os.path.join(ROOT, "profiles", b), "w").write("\n".join(lines) + "\n")
        print("profiles/" + b)
    else:
        print("missing", a)
