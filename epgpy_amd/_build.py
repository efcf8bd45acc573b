"""Build libepgx.so (hand-written HIP for gfx950) in-tree with hipcc.

The shared library is built next to its sources (epgpy_amd/csrc/libepgx.so) so that it
travels with a snapshot of the repository; it is git-ignored.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBPATH = os.path.join(CSRC, "libepgx.so")
SOURCES = ["epgx_api.hip"]
DEPENDS = ["epgx_api.hip", "epgx_kernels.hip.h", os.path.join("..", "..", "include", "epgx.h")]
ARCH = "gfx950"


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libepgx.so cannot be built")
    return exe


def needs_build():
    if not os.path.exists(LIBPATH):
        return True
    built = os.path.getmtime(LIBPATH)
    return any(os.path.getmtime(os.path.join(CSRC, d)) > built for d in DEPENDS)


def build(force=False, verbose=False):
    """compile the HIP sources into csrc/libepgx.so; returns the library path"""
    if not force and not needs_build():
        return LIBPATH
    # -structurizecfg-skip-uniform-regions: every branch of run_kernel is wave-uniform (scalar
    # compares on record flags); without this option the AMDGPU backend still structurizes the
    # record dispatch and threads it with mask registers (~20 extra SALU instructions per record)
    cmd = [hipcc(), f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-mllvm", "-structurizecfg-skip-uniform-regions=1",
           # the first 16 dwords of run_kernel's arguments are preloaded into SGPRs at wave launch
           "-mllvm", "-amdgpu-kernarg-preload-count=16", "-o", LIBPATH] + SOURCES
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return LIBPATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
