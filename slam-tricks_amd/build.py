"""Builds libstba.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting
slam-tricks_amd/libstba.so travels to the GPU box with the repository snapshot.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SOURCES = ["dense_chol.hip", "ba_kernels.hip", "stba_engine.hip", "pg_engine.hip", "small_dense.hip", "two_view.hip", "calib_io.cpp", "comm.cpp"]
HEADERS = ["common.hpp", "ba_kernels.hpp", "small_linalg.hpp", os.path.join("..", "..", "include", "stba.h")]
LIB = os.path.join(HERE, "libstba.so")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-munsafe-fp-atomics", "-Wall",
         "-Wno-unused-function", "-Wno-unused-result"]


# STBA_DEBUG_KNOBS=1 in the environment of the BUILD compiles the scheduling-experiment knobs in (environment variables read
# by dense_chol.hip; tools/mega_trace.py, tools/sim_sweep.py); the product build reads none
if os.environ.get("STBA_DEBUG_KNOBS", "0") not in ("", "0"):
    FLAGS.append("-DSTBA_DEBUG_KNOBS")


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


HEAD_FILE = os.path.join(HERE, "BUILD_HEAD")


def source_hash():
    """content hash of everything the library is compiled from (csrc + include/stba.h): identifies a BUILD independently of
    commits that only touch documents or profiles"""
    import hashlib
    h = hashlib.sha1()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp", ".cpp")))
    files.append(os.path.join(HERE, "..", "include", "stba.h"))
    for f in files:
        with open(f, "rb") as fh:
            h.update(os.path.basename(f).encode() + b"\0" + fh.read())
    return h.hexdigest()[:12]


def stamp_head():
    """git head of the tree the library was built from -> slam-tricks_amd/BUILD_HEAD (git-ignored; it travels to the GPU box,
    where there is no .git: bench.py and the tools/pmc_*.sh scripts put it into what they write, so that a counter file
    can be held against the bench line it belongs to)."""
    try:
        root = os.path.dirname(HERE)
        head = subprocess.check_output(["git", "-C", root, "rev-parse", "--short", "HEAD"], stderr=subprocess.DEVNULL, text=True).strip()
        dirty = subprocess.call(["git", "-C", root, "diff", "--quiet", "HEAD", "--", "slam-tricks_amd/csrc", "include"]) != 0
        with open(HEAD_FILE, "w") as f:
            f.write(head + ("+dirty" if dirty else "") + " src:" + source_hash() + "\n")
    except Exception:
        pass                      # no git here (the GPU box): keep the file that came with the snapshot


def build_head():
    """'<git head>[+dirty] src:<hash>' of the tree the library was built from; the src: part is recomputed from the files at hand
    (it must describe what is loaded, and a commit of documents moves the git head but not the library)"""
    try:
        head = open(HEAD_FILE).read().strip().split(" src:")[0]
    except OSError:
        head = "unknown"
    return head + " src:" + source_hash()


def build(force=False, verbose=False):
    stamp_head()
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs = []
    procs = []
    for s in SOURCES:
        o = os.path.join(CSRC, s.replace(".hip", ".o").replace(".cpp", ".o"))
        cmd = [hipcc, *FLAGS, "-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd)))
        objs.append(o)
    for cmd, p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs, "-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
