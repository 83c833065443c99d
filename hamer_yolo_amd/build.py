"""Build libhamer_hip.so for gfx950 with hipcc, in-tree (hamer_yolo_amd/libhamer_hip.so).

hipcc cross-compiles without a GPU; the built .so travels to the GPU box with the repo
snapshot (it is git-ignored, not gpurun-ignored).
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libhamer_hip.so")
SOURCES = ["status.hip", "gemm.hip", "norm.hip", "attention.hip", "patch.hip", "decoder.hip", "mano.hip", "tome.hip", "forward.hip", "prof.hip", "yolo.hip"]


def _stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "hamer_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = True, ablations: bool = False) -> str:
    """ablations=True: the timing-ablation build (-DHM_ABLATIONS: GEMM variants that skip work and give WRONG results) into
    a separate libhamer_hip_abl.so that only tools/bench_gemm_ab.py loads; the product library never contains them."""
    lib = LIB.replace(".so", "_abl.so") if ablations else LIB
    if not ablations and not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    bdir = os.path.join(HERE, "build_abl" if ablations else "build")
    os.makedirs(bdir, exist_ok=True)
    for src in SOURCES:
        obj = os.path.join(bdir, src.replace(".hip", ".o"))
        cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"] + \
              (["-DHM_ABLATIONS"] if ablations else []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    failed = False
    for src, p in procs:
        outp, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"[build] {src} FAILED\n{outp}\n")
        elif verbose and outp.strip():
            sys.stderr.write(f"[build] {src}:\n{outp}\n")
    if failed:
        raise RuntimeError("hipcc failed building libhamer_hip.so")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    if verbose:
        sys.stderr.write(f"[build] wrote {lib}\n")
    return lib


if __name__ == "__main__":
    build(force="--force" in sys.argv, ablations="--ablations" in sys.argv)
