"""
Build libbfg_mi355.so (the hand-written gfx950 kernels + C-ABI) in-tree with hipcc.
hipcc cross-compiles for gfx950 without a GPU, so this also runs in the build container.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "bfg_mi355.hip")
DEPS = sorted(os.path.join(HERE, "csrc", f) for f in os.listdir(os.path.join(HERE, "csrc"))
              if f.endswith((".hip", ".hpp"))) + [os.path.join(os.path.dirname(HERE), "include", "bfg_mi355.h")]
SO = os.path.join(HERE, "libbfg_mi355.so")

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               "-munsafe-fp-atomics",   # f64 atomicAdd -> global_atomic_add_f64 / ds_add_f64, no CAS loop
               "-ffp-contract=on"]


def find_hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: cannot build libbfg_mi355.so")


def needs_build():
    if not os.path.exists(SO):
        return True
    t = os.path.getmtime(SO)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return SO
    cmd = [find_hipcc()] + HIPCC_FLAGS + ["-o", SO, SRC]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=HERE)
    return SO


if __name__ == "__main__":
    print(build(force=True, verbose=True))
