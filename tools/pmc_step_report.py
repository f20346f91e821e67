"""Per-kernel HBM-side bandwidth and MFMA utilisation of one training step from three rocprofv3 --pmc passes
(FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES + GRBM_GUI_ACTIVE; separate runs of
`bench.py --steps 1 --warmup 1 --single-stream --no-roofline --no-cpu-baseline`, csv output with --kernel-trace).

gfx950 corrections per /opt/skills/guides/MI355X_MICROARCH.md: FETCH_SIZE / WRITE_SIZE are KiB of L2 memory-side requests
(Infinity-Cache hits included); FETCH_SIZE is doubled (wide coalesced reads are tallied at half their bytes).
MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x SIMDs) -- the SQ counter is summed over the chip (checked here
against 32 cycles x the known MFMA count of the GEMM launches), GRBM_GUI_ACTIVE (summed over the 8 XCDs, hence / 8) is the kernel's active cycles (it includes a few us around
every dispatch, so it is not used as a clock estimate).

  python tools/pmc_step_report.py <fetch.csv> <write.csv> <mfma.csv> [gemm_flops_per_step]
"""
import collections
import csv
import re
import sys


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name).replace("void ", "")
    name = re.sub(r"\(.*$", "", name)
    return name[:70]


def load(path):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    seen = set()
    with open(path) as f:
        for r in csv.DictReader(f):
            k = short(r["Kernel_Name"])
            per[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if r["Dispatch_Id"] not in seen:
                seen.add(r["Dispatch_Id"])
                dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)   # us
    return per, dur


fetch, _ = load(sys.argv[1])
write, _ = load(sys.argv[2])
mfma, dur = load(sys.argv[3])
STEPS = 2          # 1 warm-up + 1 timed step in the profiled command
SIMDS = 256 * 4
XCDS = 8           # GRBM_GUI_ACTIVE comes back summed over the 8 XCDs (per-XCD value = kernel duration x ~2.4 GHz)
rows = []
for k in dur:
    n = len(dur[k])
    us = sum(dur[k]) / n
    rd = 2.0 * 1024.0 * sum(fetch[k]["FETCH_SIZE"]) / max(1, len(fetch[k]["FETCH_SIZE"])) if k in fetch else 0.0
    wr = 1024.0 * sum(write[k]["WRITE_SIZE"]) / max(1, len(write[k]["WRITE_SIZE"])) if k in write else 0.0
    busy = sum(mfma[k]["SQ_VALU_MFMA_BUSY_CYCLES"]) / n if mfma[k]["SQ_VALU_MFMA_BUSY_CYCLES"] else 0.0
    act = sum(mfma[k]["GRBM_GUI_ACTIVE"]) / n / XCDS if mfma[k]["GRBM_GUI_ACTIVE"] else 0.0
    rows.append((sum(dur[k]) / STEPS, k, n // STEPS, us, rd, wr, busy, act))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("| kernel | launches/step | avg us | % of step kernel time | read MB | write MB | L2-memory-side GB/s | MFMA busy % |")
print("|---|---|---|---|---|---|---|---|")
for t, k, n, us, rd, wr, busy, act in rows[:28]:
    gbs = (rd + wr) / us / 1e3 if us > 0 else 0.0
    util = 100.0 * busy / (act * SIMDS) if act > 0 else 0.0
    print(f"| `{k}` | {n} | {us:.1f} | {100 * t / tot:.1f} | {rd / 1e6:.0f} | {wr / 1e6:.0f} | {gbs:.0f} | {util:.1f} |")
print(f"\nkernel time per step {tot / 1e3:.1f} ms (under the counter pass)")
gemm_busy = sum(sum(mfma[k]["SQ_VALU_MFMA_BUSY_CYCLES"]) for k in mfma if "gemm" in k) / STEPS
all_busy = sum(sum(mfma[k]["SQ_VALU_MFMA_BUSY_CYCLES"]) for k in mfma) / STEPS
all_act = sum(sum(mfma[k]["GRBM_GUI_ACTIVE"]) for k in mfma) / STEPS / XCDS
print(f"whole step: MFMA busy {100 * all_busy / (all_act * SIMDS):.1f} % of the active cycles of all SIMDs")
if len(sys.argv) > 4:
    fl = float(sys.argv[4])
    print(f"check of the counter's convention: GEMM launches {gemm_busy:.3e} busy cycles per step vs 32 x (GEMM FLOPs / 32768) = {32 * fl / 32768:.3e}")
