#!/usr/bin/env python3
"""Builds another libmagbert_hip.so under gpurun_ab/<name>/ (git-ignored, travels to the GPU box) for same-box A/B runs:
   python scripts/build_variant.py <name> [-DFLAG ...] [--files gemm.hip,...]
Only the listed files (default: gemm.hip) are recompiled with the extra flags; the other objects come from the in-tree build.
The tools have RUNPATH, so LD_LIBRARY_PATH=gpurun_ab/<name> selects the build (scripts/gpu_ab.sh, scripts/gpu_kstats.sh)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bert_multimodal_transformer_amd import build as b

def main():
    name = sys.argv[1]
    flags = [a for a in sys.argv[2:] if a.startswith("-")and not a.startswith("--files")]
    files = ["gemm.hip"]
    for a in sys.argv[2:]:
        if a.startswith("--files="):
            files = a.split("=", 1)[1].split(",")
    b.build()
    out = os.path.join(ROOT, "gpurun_ab", name)
    os.makedirs(out, exist_ok=True)
    objs = []
    for s in b.SOURCES:
        o = os.path.join(b.LIBDIR, "obj", s.replace(".hip", ".o"))
        if s in files:
            o = os.path.join(out, s.replace(".hip", ".o"))
            r = subprocess.run([b._hipcc()] + b.FLAGS + flags + ["-c", os.path.join(b.CSRC, s), "-o", o], capture_output=True, text=True)
            if r.returncode:
                sys.exit(r.stderr[-4000:])
        objs.append(o)
    lib = os.path.join(out, "libmagbert_hip.so")
    r = subprocess.run([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, capture_output=True, text=True)
    if r.returncode:
        sys.exit(r.stderr[-4000:])
    print(lib)

if __name__ == "__main__":
    main()
