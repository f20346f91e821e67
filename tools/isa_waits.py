"""Where does the compiler make a kernel wait?  Compact per-kernel trace of the gfx950 ISA: loop headers, global loads /
stores, LDS-DMA, MFMA runs, barriers and every s_waitcnt, in program order.

Reading these traces is how this round's waitcnt findings were made: `vmcnt(0)` in front of `ds_read_b64_tr_b16`
after an LDS-DMA, `lgkmcnt(0)` before the DMA issue, `vmcnt(0)` per output row in a branchy epilogue, `vmcnt(0)` right after a
prefetch whose loads sit behind guards.  A counted wait (`vmcnt(6)`) means the pass could follow the loads; `vmcnt(0)` directly
after a run of loads in a loop usually means it could not.

  python tools/isa_waits.py lightly-train_amd/csrc/gemm.hip gemm256q_kernel        # kernels whose name contains the pattern
  python tools/isa_waits.py lightly-train_amd/csrc/elementwise.hip layernorm_bwd --loop-only
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def compile_to_isa(src: str) -> str:
    out = os.path.join(tempfile.mkdtemp(prefix="lt_isa_"), "k.s")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(ROOT, "lightly-train_amd", "csrc"), "-S", "--cuda-device-only", "-o", out, src]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    return out


def token(line: str):
    t = line.strip()
    if "Loop Header" in line:
        return "LOOP[d=" + line.split("Depth=")[-1].strip() + "]"
    if t.startswith("s_waitcnt"):
        return "[" + t.replace("s_waitcnt ", "") + "]"
    if t.startswith("global_load_lds") or (t.startswith("buffer_load") and " lds" in t):
        return "DMA"
    if t.startswith(("global_load", "buffer_load", "flat_load")):
        return "ld"
    if t.startswith(("global_store", "buffer_store", "flat_store", "global_atomic")):
        return "st"
    if t.startswith("ds_read") or "ds_read" in t and t.startswith(";") is False and "ASM" not in t:
        return "r"
    if t.startswith("ds_write"):
        return "w"
    if t.startswith("v_mfma"):
        return "M"
    if t.startswith("s_barrier"):
        return "BAR"
    return None


def main() -> None:
    src, pat = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else ""
    loop_only = "--loop-only" in sys.argv
    lines = open(compile_to_isa(src)).read().split("\n")
    i = 0
    while i < len(lines):
        l = lines[i]
        if l.startswith("_Z") and "@" in l and pat in l:
            name = re.sub(r"^_ZN?\d*_GLOBAL__N_1", "", l.split(":")[0])
            j, toks, seen_loop = i, [], False
            while not lines[j].startswith(".Lfunc_end"):
                tk = token(lines[j])
                if tk is not None:
                    seen_loop = seen_loop or tk.startswith("LOOP")
                    if not loop_only or seen_loop:
                        toks.append(tk)
                j += 1
            runs = []
            for t in toks:
                if runs and runs[-1][0] == t and not t.startswith("["):
                    runs[-1][1] += 1
                else:
                    runs.append([t, 1])
            meta = {}
            for k in range(j, min(j + 400, len(lines))):
                m = re.search(r"\.(vgpr_count|vgpr_spill_count|sgpr_count):\s+(\d+)", lines[k])
                if m and m.group(1) not in meta:
                    meta[m.group(1)] = m.group(2)
            print(name[:110])
            print("   " + " ".join(f"{t}x{c}" if c > 1 else t for t, c in runs))
            i = j
        i += 1


if __name__ == "__main__":
    main()
