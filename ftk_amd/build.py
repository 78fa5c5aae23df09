"""Builds ftk_amd/libftkx.so (HIP kernels + C ABI + C++ tracker) for gfx950, in-tree.

    python -m ftk_amd.build            # hipcc cross-compiles without a GPU present

-ffp-contract=off is part of the contract, not a tuning flag: the FP64 hit path and the derived-field kernels must not fuse
a*b+c (the x86-64 reference build has no FMA contraction; the one fused op, std::fma in eigen_solver2.hh:32, is explicit)."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = [os.path.join(HERE, "csrc", f) for f in ("tile_kernels.hip", "mask_kernels.hip", "cull_exact_kernels.hip", "halo_kernels.hip", "derive_kernels.hip", "ftkx_api.hip", "prepare.hip", "collect.hip", "halo.hip", "series.hip", "series_kernels.hip", "one_kernel.hip", "dist_kernels.hip", "trace_device.hip", "tracker.cpp", "trace.cpp", "io.cpp", "slab.cpp", "slab_rccl.cpp", "upload.cpp")]
OUT = os.path.join(HERE, "libftkx.so")
FLAGS = ["-O3", "--offload-arch=gfx950", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function", "-Wno-unused-variable"]


def hipcc():
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError("hipcc not found")


def deps():
    d = list(SRC)
    for root in (os.path.join(HERE, "csrc"), os.path.join(os.path.dirname(HERE), "include")):
        d += [os.path.join(root, f) for f in os.listdir(root) if f.endswith((".hpp", ".h", ".hh"))]
    return d


def up_to_date():
    return os.path.exists(OUT) and all(os.path.getmtime(OUT) >= os.path.getmtime(p) for p in deps())


def build(force=False, verbose=False):
    """one object per source under ftk_amd/csrc/build/ (rebuilt when the source or any header is newer), then one link"""
    if not force and up_to_date():
        return OUT
    objdir = os.path.join(HERE, "csrc", "build")
    os.makedirs(objdir, exist_ok=True)
    headers = [p for p in deps() if p not in SRC]
    newest_header = max(os.path.getmtime(h) for h in headers)
    cflags = [f for f in FLAGS if f != "-shared"] + os.environ.get("FTKX_EXTRA_CFLAGS", "").split()     # (diagnostic builds: -DFTKX_TILE_STAMPS)
    objs, procs = [], []
    for src in SRC:
        obj = os.path.join(objdir, os.path.basename(src) + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), newest_header):
            cmd = [hipcc()] + cflags + ["-c", src, "-o", obj]
            # the tile kernels' register / scratch / occupancy figures go to a file next to the object: tests/test_kernel_budget.py holds
            # them to their budget without compiling the file a second time
            keep = os.path.basename(src) == "tile_kernels.hip"
            if verbose or keep:
                cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
            if verbose:
                print(" ".join(cmd))
            err = open(obj + ".resources.txt", "w") if keep and not verbose else None
            procs.append((cmd, subprocess.Popen(cmd, stderr=err), err))
    for cmd, p, err in procs:
        rc = p.wait()
        if err:
            err.close()
        if rc != 0:
            if err:
                sys.stderr.write(open(err.name).read()[-4000:])
            raise subprocess.CalledProcessError(rc, cmd)
    subprocess.check_call([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs + ["-lrccl"])     # (slab_rccl.cpp: the slab pass's messages over RCCL)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose="-v" in sys.argv)
    print(OUT)
