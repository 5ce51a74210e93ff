"""Builds libmagbert_hip.so (gfx950) in-tree with hipcc.  No torch C++ extension: the product boundary is a plain
C ABI (include/magbert_hip.h) loaded with ctypes, so the build is `hipcc -c` per kernel file + one link."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libmagbert_hip.so")
SOURCES = ["gemm.hip", "gemm_pp.hip", "rowops.hip", "mag.hip", "attention.hip", "xlnet_attention.hip", "xlnet_rowops.hip", "head.hip", "adamw.hip",
           "engine.hip", "xlnet_engine.hip", "comm.hip"]
HEADERS = ["common.h", "kernels.h", "gemm_tile.h", "adamw_dev.h", "attn_common.h", "engine_common.h", "comm.h", "mag_pack.h", os.path.join("..", "..", "include", "magbert_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-fPIC", "-fvisibility=default",
         "-Wno-unused-result"]


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    jobs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        if force or _stale(obj, [src] + hdrs):
            jobs.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [_hipcc()] + FLAGS + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr[-4000:]))
        return obj

    if jobs:
        if verbose:
            print("[magbert build] compiling %d file(s) for gfx950" % len(jobs), file=sys.stderr)
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(cc, jobs))
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(LIB, objs):
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s" % r.stderr[-4000:])
    return LIB


def build_tools(force=False):
    """tools/*.cpp: torch-free C++ drivers of the C ABI (measurement tooling): step_bench (whole optimizer steps), gemm_bench."""
    root = os.path.dirname(HERE)
    outdir = os.path.join(root, "tools", "bin")
    outs = []
    for name in ("step_bench", "gemm_bench", "launch_floor", "attn_bench", "adamw_bench", "event_capture_probe", "stream_handoff_probe", "mfma_lds_probe", "atomic_probe"):
        src = os.path.join(root, "tools", name + ".cpp")
        if not os.path.exists(src):
            continue
        os.makedirs(outdir, exist_ok=True)
        out = os.path.join(outdir, name)
        if force or _stale(out, [src, LIB, os.path.join(root, "include", "magbert_hip.h")]):
            cmd = [_hipcc(), "--offload-arch=gfx950", "-O2", "-std=c++17", src, "-o", out, "-L" + LIBDIR, "-lmagbert_hip",
                   "-Wl,-rpath,$ORIGIN/../../bert_multimodal_transformer_amd/lib"]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr[-4000:]))
        outs.append(out)
    return outs


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
    print(build_tools(force="--force" in sys.argv))
