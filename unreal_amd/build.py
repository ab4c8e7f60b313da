"""Build libunreal_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libunreal_hip.so")


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    srcs = glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.h"))
    return any(os.path.getmtime(s) > t for s in srcs)


def build_library(force=False, verbose=True):
    os.makedirs(LIB_DIR, exist_ok=True)
    if not force and not needs_build():
        return LIB_PATH
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    objs = []
    procs = []
    for s in srcs:
        o = os.path.join(LIB_DIR, os.path.basename(s)[:-4] + ".o")
        objs.append(o)
        cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17"] + os.environ.get("UNREAL_HIPCC_FLAGS", "").split() + \
              ["-c", s, "-o", o]
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), out.decode()))
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("link failed: %s\n%s" % (" ".join(cmd), r.stdout.decode()))
    if verbose:
        print("built", LIB_PATH, os.path.getsize(LIB_PATH), "bytes", file=sys.stderr)
    return LIB_PATH


if __name__ == "__main__":
    build_library(force="--force" in sys.argv)
