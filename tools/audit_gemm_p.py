"""Audit of the persistent GEMM's code object (gemm_p.hip hand-allocates the accumulator file: DESIGN 4.1b): every AGPR reference must sit
inside an inline-asm block, no scratch, 256 AGPRs in the descriptor.  Usage: python tools/audit_gemm_p.py [file.s]  (default: compile)."""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def compile_asm() -> str:
    d = tempfile.mkdtemp()
    src = os.path.join(ROOT, "lightly-train_amd", "csrc", "gemm_p.hip")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-result",
                    "-x", "hip", "-S", "--cuda-device-only", src, "-o", os.path.join(d, "gemm_p.s")], check=True, cwd=d)
    return os.path.join(d, "gemm_p.s")


def audit(path: str) -> dict:
    txt = open(path).read()
    out = {}
    for m in re.finditer(r"\n(_ZN7lt_gemm\S*gemm1p_kernel\S+):[^\n]*\n(.*?)\n\s+\.end_amdhsa_kernel", txt, re.S):
        name, body = m.group(1), m.group(2)
        in_asm, bad, mfma, scratch = False, [], 0, 0
        for ln in body.split("\n"):
            if "#ASMSTART" in ln:
                in_asm = True; continue
            if "#ASMEND" in ln:
                in_asm = False; continue
            code = ln.split(";")[0]
            if "v_mfma" in code:
                mfma += 1
            if "scratch_" in code:
                scratch += 1
            if not in_asm and re.search(r"\ba(\d+|\[\d+(:\d+)?\])", code) and not code.strip().startswith("."):
                bad.append(ln.strip())
        agpr = re.search(r"\.amdhsa_accum_offset (\d+)", body)
        nfree = re.search(r"\.amdhsa_next_free_vgpr (\d+)", body)
        out[name] = dict(mfma=mfma, scratch=scratch, agpr_outside_asm=bad[:5], n_bad=len(bad), accum_offset=int(agpr.group(1)) if agpr else None,
                         next_free_vgpr=int(nfree.group(1)) if nfree else None)
    return out


if __name__ == "__main__":
    res = audit(sys.argv[1] if len(sys.argv) > 1 else compile_asm())
    ok = True
    for k, v in res.items():
        good = v["n_bad"] == 0 and v["scratch"] == 0 and v["next_free_vgpr"] is not None and v["next_free_vgpr"] - v["accum_offset"] == 256
        ok &= good
        print(("ok  " if good else "BAD ") + k, v)
    sys.exit(0 if ok and res else 1)
