"""In-tree build of the native extension ``distributed_pytorch_amd/_C*.so``.

``python -m distributed_pytorch_amd.build`` compiles every ``csrc/kernels/*.hip`` for gfx950 with
hipcc (cross-compiles without a GPU), the C++ runtime / binding, and links one pybind11 module that
sits next to the package so it travels to the GPU box with the tree.  Objects are cached under
``build/`` and rebuilt only when a source (or any header) is newer.
"""
from __future__ import annotations

import os
import subprocess
import sys
import sysconfig
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
HIPCC = os.path.join(ROCM, "bin", "hipcc")
ARCH = "gfx950"


def _sources():
    out = []
    for root, _, files in os.walk(CSRC):
        for f in sorted(files):
            if f.endswith((".hip", ".cpp")):
                out.append(os.path.join(root, f))
    return out


def _headers_mtime() -> float:
    m = 0.0
    for root, _, files in os.walk(CSRC):
        for f in files:
            if f.endswith(".h"):
                m = max(m, os.path.getmtime(os.path.join(root, f)))
    return m


def target_path() -> str:
    return os.path.join(HERE, "_C" + sysconfig.get_config_var("EXT_SUFFIX"))


def build(verbose: bool = False, force: bool = False, jobs: int = 0) -> str:
    import pybind11

    objdir = os.path.join(os.path.dirname(HERE), "build", "obj")
    os.makedirs(objdir, exist_ok=True)
    inc = ["-I" + pybind11.get_include(), "-I" + sysconfig.get_paths()["include"], "-I" + CSRC,
           "-I" + os.path.join(ROCM, "include")]
    common = [HIPCC, "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
              "-Wno-unused-result"] + inc + os.environ.get("DPA_BUILD_DEFS", "").split()
    hmt = _headers_mtime()
    todo, objs = [], []
    for src in _sources():
        obj = os.path.join(objdir, os.path.relpath(src, CSRC).replace(os.sep, "_") + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hmt):
            todo.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = common + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("compile failed: %s\n%s" % (src, r.stderr[-4000:]))
        return obj

    jobs = jobs or min(8, int(os.environ.get("MAX_JOBS", "8")), max(1, len(todo)))
    if todo:
        with ThreadPoolExecutor(max_workers=jobs) as ex:
            list(ex.map(compile_one, todo))
    tgt = target_path()
    if todo or not os.path.exists(tgt):
        # no rpath: at run time libamdhip64.so.7 resolves to the copy torch loaded first
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", tgt] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s" % r.stderr[-4000:])
    return tgt


if __name__ == "__main__":
    path = build(verbose="-v" in sys.argv, force="-f" in sys.argv)
    print("built", path)
