"""Build libmft_hip.so (hand-written HIP kernels, gfx950) in-tree with hipcc.

    python -m meta_fine_tuning_amd.build      (or __graft_entry__.build())
"""
import glob
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
LIB = os.path.join(PKG, "libmft_hip.so")


def sources():
    return sorted(glob.glob(os.path.join(PKG, "csrc", "*.hip")))


def _flags_stamp():
    return "experiments" if os.environ.get("MFT_EXPERIMENTS", "0") == "1" else "product"


def needs_build():
    if not os.path.exists(LIB):
        return True
    stamp = os.path.join(PKG, "csrc", "_obj", "flags.txt")
    if not os.path.exists(stamp) or open(stamp).read().strip() != _flags_stamp():
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(PKG, "csrc", "*.h")) + [os.path.join(ROOT, "include", "mft_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True, jobs=None):
    """One object per .hip source (recompiled only when it or a header is newer), compiled ``jobs`` at a time, then linked."""
    if not force and not needs_build():
        return LIB
    stamp = os.path.join(PKG, "csrc", "_obj", "flags.txt")
    if os.path.exists(stamp) and open(stamp).read().strip() != _flags_stamp():
        force = True                              # product <-> experiments: every object is rebuilt
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(PKG, "csrc", "_obj")
    os.makedirs(objdir, exist_ok=True)
    hdrs = glob.glob(os.path.join(PKG, "csrc", "*.h")) + [os.path.join(ROOT, "include", "mft_hip.h")]
    if _flags_stamp() == "experiments":           # the measured-slower variants live outside the package and are #included under the flag
        hdrs += glob.glob(os.path.join(ROOT, "tools", "experiments", "*.inc"))
    hdr_t = max(os.path.getmtime(h) for h in hdrs)
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
             "-I" + os.path.join(PKG, "csrc")]
    if _flags_stamp() == "experiments":           # MFT_EXPERIMENTS=1: also the measured-slower kernel variants and ablation aids (tools/)
        flags.append("-DMFT_EXPERIMENTS")
    todo, objs = [], []
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
            todo.append([hipcc] + flags + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)

    jobs = jobs or int(os.environ.get("MFT_BUILD_JOBS", "4"))
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        list(ex.map(run, todo))
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB])
    with open(os.path.join(objdir, "flags.txt"), "w") as f:
        f.write(_flags_stamp() + "\n")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
