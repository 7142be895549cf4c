"""Build-time guard for the 8-phase GEMM kernels (csrc/gemm8p.hip): they live at the 256-VGPR edge, and a scratch access INSIDE the K loop is a
catastrophe there -- a scratch_load is a VMEM load, the compiler follows it with s_waitcnt vmcnt(0), and that drains the LDS-DMA queue the loop
keeps two K tiles deep (measured: the 320-row forward kernel 1.58 -> 1.91 ms per step from TWO such reloads).  Compiles the file with -save-temps
and fails if any gemm8 kernel has a scratch instruction between its K-loop header and the loop's backward branch -- or, round 6, if a 256-row
forward kernel without per-tensor scales (plain / SwiGLU / QKV / MX epilogues: the ones whose epilogue makes a pass per 8 or 32 rows) has ANY scratch
access: a reload inside such a pass is followed by s_waitcnt vmcnt(0), i.e. by a wait for the stores the pass has just issued (the MX SwiGLU launch of
the mxfp8 sampler: 863 -> 755 us when its 64 reloads went away, profiles/r06_epilogue_waits.txt).
    python tools/check_spills.py [extra hipcc flags]           (needs hipcc; ~10 s; run by tests/test_oracle_cpu.py)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRCS = [os.path.join(ROOT, "stable-diffusion-3-from-scratch_amd", "csrc", f) for f in ("gemm8p.hip", "gemm8p_inf.hip")]      # the two translation units of the kernel
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")      # (the variable build.py honours)
NO_COMPILER = 77                                            # exit code: hipcc not found (callers skip)


def main():
    import shutil
    if shutil.which(HIPCC) is None:
        print(f"check_spills: compiler {HIPCC!r} not found (set HIPCC)", file=sys.stderr)
        return NO_COMPILER
    asm = []
    for src in SRCS:
        stem = os.path.basename(src)[:-4]
        with tempfile.TemporaryDirectory() as td:
            r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-save-temps"] + sys.argv[1:] + ["-c", src, "-o", os.path.join(td, "o.o")],
                               cwd=td, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)
            if r.returncode != 0:
                print(r.stderr[-4000:], file=sys.stderr)
                return 2
            asm += open(os.path.join(td, stem + "-hip-amdgcn-amd-amdhsa-gfx950.s")).read().split("\n")
    kernels, bad = audit(asm)
    print(f"{kernels} gemm8 kernels checked; in-loop scratch / scratch in the no-scratch kernels / copied claim registers: {bad if bad else 'none'}")
    return 1 if bad or not kernels else 0


def audit(asm):
    """(number of gemm8 kernels found, list of (kernel, finding)) for the lines of a gfx950 .s file (tests/test_oracle_cpu.py feeds it synthetic kernels)."""
    bad, kernels = [], 0
    i = 0
    while i < len(asm):
        m = re.match(r"^(_ZN12_GLOBAL__N_1\d+gemm8\w*_kernel\w+):", asm[i])
        if not m:
            i += 1
            continue
        name, j = m.group(1), i + 1
        while j < len(asm) and "s_endpgm" not in asm[j]:
            j += 1
        body = asm[i:j]
        kernels += 1
        # dynamic tile claiming: the returning atomic is issued from inline asm by one lane and its destination register is consumed behind an explicit
        # s_waitcnt vmcnt(0) much later (csrc/gemm8p.hip claim_issue / claim_take) -- the compiler does not know the register is filled asynchronously,
        # so it must never COPY, SPILL or OVERWRITE it while a claim may be in flight: after the first claim the register may only be written by claim
        # atomics and read by ordinary instructions (found in round 6: the per-tensor fp8 SwiGLU kernel moved it through v_mov pairs -- wrong tiles)
        claims = [k for k, l in enumerate(body) if "global_atomic_add" in l and " sc0" in l and any("s_mov_b64 exec, 1" in body[x] for x in range(max(0, k - 3), k))]
        for reg in sorted({int(re.search(r"global_atomic_add v(\d+)", body[k]).group(1)) for k in claims}):
            first = min(k for k in claims if re.search(rf"global_atomic_add v{reg}\b", body[k]))

            def has(tok):
                m = re.fullmatch(r"v(\d+)", tok)
                if m:
                    return int(m.group(1)) == reg
                m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
                return bool(m) and int(m.group(1)) <= reg <= int(m.group(2))

            for k in range(first + 1, len(body)):
                t = body[k].split(";")[0].strip()
                if not t or t.endswith(":") or k in claims:
                    continue
                op, _, rest = t.partition(" ")
                toks = [x.strip() for x in rest.replace("|", "").split(",")]
                toks = [x.split(" ")[0] for x in toks if x]
                if not any(has(x) for x in toks):
                    continue
                writes = bool(toks) and has(toks[0]) and not op.startswith(("v_cmp", "global_store", "ds_write", "scratch_store", "buffer_store", "s_"))
                if writes or op.startswith(("v_mov", "v_accvgpr", "scratch_store", "v_swap", "v_permlane")):
                    bad.append((name, f"claim register v{reg} copied / overwritten: {t}"))
                    break
        # <256 rows, !WGRAD, !KMAJOR-B, epilogue, ..., per-tensor scales = false>: no scratch at all
        if re.search(r"gemm8_kernelILi256ELb0ELb0ELi\d+ELb0ELb[01]ELb[01]ELb0EEE", name):
            n = sum("scratch_" in l for l in body)
            if n:
                bad.append((name, f"{n} scratch accesses in a kernel that must have none"))
        # the K loop: the innermost loop (Depth=2) of the item loop; from its header label to the conditional branch back to that label
        for k, line in enumerate(body):
            if "Inner Loop Header: Depth=2" in line:
                label = None
                for b in range(k, max(k - 3, 0), -1):
                    mm = re.match(r"^(\.LBB\d+_\d+):", body[b])
                    if mm:
                        label = mm.group(1)
                        break
                if label is None:
                    continue
                end = next((e for e in range(len(body) - 1, k, -1) if re.search(r"s_c?branch\w* " + re.escape(label) + r"\b", body[e])), None)
                if end is None:
                    continue
                n = sum("scratch_" in l for l in body[k:end])
                if n and any("v_mfma" in l for l in body[k:end]):
                    bad.append((name, n))
        i = j
    return kernels, bad


if __name__ == "__main__":
    sys.exit(main())
