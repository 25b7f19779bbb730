"""Builds libtpp_xsmm_runner_utils.so (the drop-in for the reference's library of the
same name, runtime/Xsmm/CMakeLists.txt:1-11) with hipcc for gfx950, in-tree.

hipcc cross-compiles without a GPU. The built .so is git-ignored but travels with
the repo snapshot to the GPU box, so nothing is compiled there.
"""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO_NAME = "libtpp_xsmm_runner_utils.so"
SO_PATH = os.path.join(HERE, SO_NAME)
SOURCES = ["runtime.cpp", "host_cache.cpp", "brgemm_f32.hip", "brgemm_f32_lw.hip", "brgemm_f32_lw16.hip", "brgemm_bf16.hip", "brgemm_bf16_dma256.hip", "brgemm_bf16_small.hip", "brgemm_bf16_lw.hip", "eltwise.hip", "peer_gather.hip"]
HEADERS = ["xsmm_desc.h", "host_cache.h", "rt_core.h", "rt_mirror.h", "rt_registry.h", "rt_operands.h", "rt_tile_queue.h", "rt_scheduler.h", "rt_enqueue.h", "rt_rewrites.h", "rt_invoke.h", "rt_chain.h", "gemm_common.h", "chain_args.h", "split_scratch.h", os.path.join("..", "..", "include", "tpp_xsmm_abi.h")]
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found; cannot build the gfx950 runtime")
    return exe


def _stale():
    if not os.path.exists(SO_PATH):
        return True
    t = os.path.getmtime(SO_PATH)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """compile every HIP translation unit for gfx950 and link the shared library"""
    if not force and not _stale():
        return SO_PATH
    cc = hipcc()
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)

    def compile_one(src):
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        cmd = [cc] + FLAGS + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, r.stderr))
        return obj

    with ThreadPoolExecutor(max_workers=len(SOURCES)) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    cmd = [cc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-pthread", "-o", SO_PATH] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stderr)
    return SO_PATH


ABLATION_SOURCES = ["runtime.cpp", "host_cache.cpp", "brgemm_f32.hip", "brgemm_bf16_lw.hip"]  # the translation units that look at TPP_HIP_ABLATION


def build_side(name, defs, sources=None, verbose=False):
    """A side build of the library into tools/<name>/ with extra -D switches on the translation units `sources` (default: the ones
    that look at TPP_HIP_ABLATION); every other object comes from the product build. For A/B and timing experiments only: use it
    through `LD_LIBRARY_PATH=tools/<name> tools/mlp_probe ...` or `TPP_XSMM_LIBRARY=tools/<name>/libtpp_xsmm_runner_utils.so
    python ...`; nothing in the package, the tests or bench.py loads it."""
    build()
    cc = hipcc()
    root = os.path.dirname(HERE)
    sources = list(sources or ABLATION_SOURCES)
    outdir = os.path.join(root, "tools", name)
    objdir = os.path.join(HERE, "build", name)
    os.makedirs(outdir, exist_ok=True)
    os.makedirs(objdir, exist_ok=True)

    def compile_one(src):
        obj = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
        cmd = [cc] + FLAGS + list(defs) + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, r.stderr))
        return obj

    with ThreadPoolExecutor(max_workers=len(sources)) as ex:
        objs = list(ex.map(compile_one, sources))
    objs += [os.path.join(HERE, "build", os.path.splitext(f)[0] + ".o") for f in SOURCES if f not in sources]
    out = os.path.join(outdir, SO_NAME)
    r = subprocess.run([cc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-pthread", "-o", out] + objs, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stderr)
    return out


def build_ablation(skip_math=False, verbose=False):
    """Side build for timing experiments: the same library with -DTPP_HIP_ABLATION (then, and only then, the loader-wave
    bf16 kernels obey TPP_HIP_CHAIN_DBG - chain_args.h; several of its bits give wrong results by design) into
    tools/_abl/ (skip_math: also -DTPP_BLW_SKIP_MATH, into tools/_abl_nomath/)."""
    defs = ["-DTPP_HIP_ABLATION"] + (["-DTPP_BLW_SKIP_MATH"] if skip_math else [])
    return build_side("_abl_nomath" if skip_math else "_abl", defs, verbose=verbose)


def build_tools(verbose=False):
    """native harness tools that link the .so: tools/tpp_replay (the stand-in for tpp-run's timing loop on
    this path) and tools/c2_probe (per-launch timing of C2 + the target of bench.py's rocprofv3 PMC passes)"""
    root = os.path.dirname(HERE)
    outs = []
    for name, extra in (("tpp_replay", ["-fopenmp"]), ("c2_probe", []), ("mlp_probe", [])):
        src = os.path.join(root, "tools", name + ".cpp")
        out = os.path.join(root, "tools", name)
        if not os.path.exists(src):
            continue
        if not (os.path.exists(out) and os.path.getmtime(out) > max(os.path.getmtime(src), os.path.getmtime(SO_PATH))):
            cmd = [hipcc(), "-O2", "-std=c++17"] + extra + [src, "-o", out, "-L", HERE, "-ltpp_xsmm_runner_utils",
                   "-Wl,-rpath,$ORIGIN/../tpp-mlir_amd", "-Wl,-rpath,/opt/rocm/lib/llvm/lib"]
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("building %s failed:\n%s" % (name, r.stderr))
        outs.append(out)
    # test helper (tests/test_chain_starved_gpu.py): a kernel that holds compute units; never loaded by the product
    src = os.path.join(root, "tools", "ubench", "cu_hog.hip")
    out = os.path.join(root, "tools", "cu_hog.so")
    if os.path.exists(src):
        if not (os.path.exists(out) and os.path.getmtime(out) > os.path.getmtime(src)):
            r = subprocess.run([hipcc(), "--offload-arch=" + ARCH, "-O2", "-shared", "-fPIC", src, "-o", out], capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("building cu_hog.so failed:\n%s" % r.stderr)
        outs.append(out)
    return outs


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    print(build_tools(verbose=True))
    if "--ablation" in sys.argv:
        print(build_ablation(verbose=True))
        print(build_ablation(skip_math=True, verbose=True))
    for a in sys.argv[1:]:  # --side=name:-DX=1,-DY  (any number of them)
        if a.startswith("--side="):
            nm, _, ds = a[len("--side="):].partition(":")
            print(build_side(nm, [d for d in ds.split(",") if d], sources=["brgemm_bf16_lw.hip"], verbose=True))
