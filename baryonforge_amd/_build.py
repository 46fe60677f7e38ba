"""
Build libbfg_mi355.so (the hand-written gfx950 kernels + C-ABI) in-tree with hipcc.
hipcc cross-compiles for gfx950 without a GPU, so this also runs in the build container.
"""
import fcntl
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
    """Compile under an exclusive file lock into a temporary file and move it into place atomically: the ranks of a
    torchrun job that all find the library stale neither interleave their linkers' writes nor load a half-written file
    (the first rank builds, the others find it fresh once they get the lock)."""
    if not force and not needs_build():
        return SO
    with open(SO + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not needs_build():
                return SO
            tmp = f"{SO}.{os.getpid()}.tmp"
            cmd = [find_hipcc()] + HIPCC_FLAGS + ["-o", tmp, SRC]
            if verbose:
                print(" ".join(cmd))
            try:
                subprocess.check_call(cmd, cwd=HERE)
                os.replace(tmp, SO)
            finally:
                if os.path.exists(tmp):
                    os.remove(tmp)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return SO


if __name__ == "__main__":
    print(build(force=True, verbose=True))
