"""Builds libcge_amd.so (hand-written HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so travels
to the GPU box with the repo snapshot (it is git-ignored, not gpurun-ignored).
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libcge_amd.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -fno-strict-aliasing: the obs staging code writes an LDS tile through float pointers and streams it out through uint32_t / uint4
# ones; under the type-based aliasing rules those float stores are dead, and hipcc did drop them (round 3, traffic staging)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
         "-Wno-unused-local-typedef", "-fno-fast-math", "-ffp-contract=off", "-fno-strict-aliasing"]


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


# Debug builds (extra_flags): -DCGE_MFG_GUARD bounds-checks every table index of the manufacturing kernel, -DCGE_GUARD the ring / draw-window /
# work-list indices of the hospital and fleet kernels (cge_device.hpp: CGE_GX); the first violation is recorded instead of dereferenced (`python -m custom_gymnasium_environments_amd.build --guard` writes libcge_amd_guard.so next to
# the release library; tools/probes/guard_run.py replays the reference fixtures through it).  -DCGE_<ENV>_TIMING: on-device phase clocks.
GUARD_FLAGS = ("-DCGE_MFG_GUARD", "-DCGE_GUARD")


def build_native(force=False, verbose=False, extra_flags=(), out=None):
    """Compile every .hip under csrc/ into one shared library.  Returns the library path."""
    if out is not None:
        return _build(out, "build_" + os.path.splitext(os.path.basename(out))[0], verbose, extra_flags)
    if not force and not is_stale():
        return LIB
    return _build(LIB, "build", verbose, extra_flags)


def _build(lib, objdir, verbose, extra_flags):
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, objdir), exist_ok=True)
    for src in sources():
        obj = os.path.join(HERE, objdir, os.path.basename(src) + ".o")
        cmd = [HIPCC, *[f for f in FLAGS if f != "-shared"], *extra_flags, "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out)
            raise RuntimeError(f"hipcc failed on {src}")
        if verbose and out.strip():
            print(out)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", lib]
    subprocess.run(cmd, check=True)
    return lib


if __name__ == "__main__":
    if "--guard" in sys.argv:
        print(build_native(verbose=True, extra_flags=GUARD_FLAGS, out=os.path.join(HERE, "libcge_amd_guard.so")))
    else:
        print(build_native(force="--force" in sys.argv, verbose=True))
