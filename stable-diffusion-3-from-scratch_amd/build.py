"""Build libmmdit_hip.so (gfx950) in-tree with hipcc.  No torch involvement: the
library is a plain C-ABI shared object (include/mmdit_hip.h).

Up-to-date checks are by CONTENT, not by mtime: every object records the sha256 of its source, of every header it can include
and of the compiler flags (csrc/<name>.o.sha), the library the hashes of its objects (libmmdit_hip.so.sha).  build(force=True)
(what __graft_entry__.build() calls) recompiles everything regardless."""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmmdit_hip.so")
HEADER = os.path.join(HERE, "..", "include", "mmdit_hip.h")
SOURCES = ["gemm.hip", "gemm_dma.hip", "gemm_lean.hip", "gemm8p.hip", "gemm8p_inf.hip", "rowops.hip", "attention.hip", "vae.hip", "optim.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + os.environ.get("MMDIT_EXTRA_HIPCC_FLAGS", "").split()


def _sha(paths, extra=""):
    h = hashlib.sha256(extra.encode())
    for p in paths:
        with open(p, "rb") as f:
            h.update(os.path.basename(p).encode() + b"\0" + f.read() + b"\0")
    return h.hexdigest()


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return None


def source_hash(src):
    """Content hash of everything object `src` depends on."""
    headers = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + [HEADER]
    if src == "gemm8p_inf.hip":
        headers.append(os.path.join(CSRC, "gemm8p.hip"))      # (it is that file, compiled with MMDIT_G8_PART 2)
    return _sha([os.path.join(CSRC, src)] + headers, " ".join([HIPCC] + FLAGS))


def build(force: bool = False, verbose: bool = False) -> str:
    want = {src: source_hash(src) for src in SOURCES}

    def cc(src):
        obj = os.path.join(CSRC, src.replace(".hip", ".o"))
        if not force and os.path.exists(obj) and _read(obj + ".sha") == want[src]:
            return obj, False
        cmd = [HIPCC] + FLAGS + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        with open(obj + ".sha", "w") as f:
            f.write(want[src])
        return obj, True

    with ThreadPoolExecutor(max_workers=8) as ex:
        res = list(ex.map(cc, SOURCES))
    objs = [o for o, _ in res]
    libsha = hashlib.sha256("".join(want[s] for s in SOURCES).encode()).hexdigest()
    if force or any(c for _, c in res) or not os.path.exists(LIB) or _read(LIB + ".sha") != libsha:
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
        with open(LIB + ".sha", "w") as f:
            f.write(libsha)
    return LIB


def is_current() -> bool:
    """True when the in-tree library was built from exactly the sources that are in the tree now."""
    libsha = hashlib.sha256("".join(source_hash(s) for s in SOURCES).encode()).hexdigest()
    return os.path.exists(LIB) and _read(LIB + ".sha") == libsha


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
