"""Build libwayverb_amd.so (HIP kernels + C ABI) for gfx950, in-tree.

`python -m wayverb_amd.build` or `wayverb_amd.build.build()`.  hipcc cross-compiles without a
GPU; the resulting .so is git-ignored but travels to the GPU box with the tree.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libwayverb_amd.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

SOURCES = ["engine.hip", "mesh_setup.hip", "node_inside.hip", "boundary_surfaces.hip", "scene_mesh.hip", "comm.cpp", "box_mesh.cpp", "filter_design.cpp", "postprocess.cpp"]
HEADERS = ["device_common.hip.h", "stream_kernels.hip.h", "boundary_kernels.hip.h", "pair_kernels.hip.h", "plane_kernels.hip.h", "triple_kernels.hip.h", "comm.h", "engine_base.h", "engine.hip.h",
           "engine_setup.hip.h", "engine_single.hip.h", "engine_pair.hip.h", "engine_triple.hip.h", "engine_batch.hip.h", "engine_io.hip.h", "engine_slab.hip.h",
           os.path.join("..", "..", "include", "wayverb_amd.h")]

# -ffp-contract=off: results must not depend on where the compiler chooses to fuse a*b+c
# (SURVEY.md Appendix A); correctly rounded fp32 divide/sqrt for the fp32-compat mode.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-fast-math", "-Wall", "-Wno-unused-function"]


RESOURCES = os.path.join(CSRC, "engine.resources.txt")  # the compiler's per-kernel register / scratch report for engine.hip


def _stale():
    if not os.path.exists(LIB) or not os.path.exists(RESOURCES):
        return True
    t = os.path.getmtime(LIB)
    files = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(f) > t for f in files)


def build(force=False, verbose=True):
    if not force and not _stale():
        return LIB
    objs = []
    for src in SOURCES:
        obj = os.path.join(CSRC, os.path.splitext(src)[0] + ".o")
        cmd = [HIPCC] + FLAGS + ["-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print("[wayverb_amd.build]", " ".join(cmd))
        if src == "engine.hip":
            # the hot kernels live at the edge of the register file (the march: 255 of 256 VGPRs): keep the compiler's
            # account of every kernel beside the library, tests/test_abi_and_host.py reads it (no kernel may spill)
            out = subprocess.run(cmd + ["-Rpass-analysis=kernel-resource-usage"], stderr=subprocess.PIPE, text=True)
            remarks = [l for l in out.stderr.splitlines() if "kernel-resource-usage" in l]
            other = [l for l in out.stderr.splitlines() if "kernel-resource-usage" not in l and not l.lstrip().startswith(("|", "^"))
                     and "__global__" not in l]
            if out.returncode != 0:
                sys.stderr.write(out.stderr)
                raise subprocess.CalledProcessError(out.returncode, cmd)
            if other and verbose:
                sys.stderr.write("\n".join(other) + "\n")
            with open(RESOURCES, "w") as f:
                f.write("\n".join(remarks) + "\n")
        else:
            subprocess.check_call(cmd)
        objs.append(obj)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB, "-ldl", "-lpthread"]
    if verbose:
        print("[wayverb_amd.build]", " ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


SANITIZED_DIR = os.path.join(HERE, "sanitized")


def build_sanitized(verbose=True):
    """The same library with its HOST code under AddressSanitizer + UndefinedBehaviorSanitizer (device code as shipped:
    -fno-gpu-sanitize), as wayverb_amd/sanitized/libwayverb_amd.so.  Test infrastructure: tools/sanitizer_run.sh puts it in
    the product library's place on a GPU box and runs the C / C++ / Python callers against it."""
    os.makedirs(SANITIZED_DIR, exist_ok=True)
    extra = ["-fsanitize=address,undefined", "-fno-gpu-sanitize", "-fno-omit-frame-pointer", "-g"]
    objs = []
    for src in SOURCES:
        obj = os.path.join(SANITIZED_DIR, os.path.splitext(src)[0] + ".o")
        cmd = [HIPCC] + FLAGS + extra + ["-x", "hip", "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print("[wayverb_amd.build]", " ".join(cmd))
        subprocess.check_call(cmd)
        objs.append(obj)
    lib = os.path.join(SANITIZED_DIR, "libwayverb_amd.so")
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-fsanitize=address,undefined", "-shared-libsan"] + objs + ["-o", lib, "-ldl", "-lpthread"]
    if verbose:
        print("[wayverb_amd.build]", " ".join(cmd))
    subprocess.check_call(cmd)
    return lib


if __name__ == "__main__":
    if "--sanitized" in sys.argv:
        print(build_sanitized())
    else:
        build(force="--force" in sys.argv)
