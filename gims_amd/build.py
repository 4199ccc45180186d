"""Build libgims_hip.so (the C-ABI kernel library, include/gims_hip.h) in-tree with hipcc for gfx950."""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libgims_hip.so")
SOURCES = ["linear.hip", "linear6.hip", "carhynet.hip", "attention.hip", "sinkhorn.hip", "sinkhorn2d.hip", "misc.hip", "agc.hip", "eval.hip", "patches.hip", "train.hip", "train_attn.hip", "optim.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-result"]
# per-file extras: keep MFMA accumulators in VGPRs (gfx950 has a unified VGPR/AGPR file) -- the softmax reads S
# straight out of the MFMA result registers instead of through v_accvgpr_read copies
# sinkhorn.hip: no SLP packing -- v_pk_*_f32 wants aligned register pairs and, next to the 192 registers the resident
# Sinkhorn kernel pins for its slab of the transport matrix, the pairing copies push that kernel into scratch spills
EXTRA = {"attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"], "train_attn.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"], "sinkhorn.hip": ["-fno-slp-vectorize"], "sinkhorn2d.hip": ["-fno-slp-vectorize"]}


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "gims_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build_lib(force: bool = False, verbose: bool = False) -> str:
    """Compile every HIP source into one shared object; returns its path."""
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for src in SOURCES:
        obj = os.path.join(HERE, "build", src.replace(".hip", ".o"))
        cmd = [hipcc, *FLAGS, *EXTRA.get(src, []), "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out, file=sys.stderr)
    tmp = f"{LIB}.{os.getpid()}.tmp"          # link next to the target, then rename: a reader never sees a half-written library
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", tmp]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    os.replace(tmp, LIB)
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv, verbose=True))
