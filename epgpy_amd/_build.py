"""Build libepgx.so (hand-written HIP for gfx950) in-tree with hipcc.

The shared library is built next to its sources (epgpy_amd/csrc/libepgx.so) so that it
travels with a snapshot of the repository; it is git-ignored.  The kernel template
instantiations are spread over several translation units that compile in parallel.
"""
import concurrent.futures
import os
import re
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJDIR = os.path.join(CSRC, "build")
LIBPATH = os.path.join(CSRC, "libepgx.so")
# (object name, source, extra flags)
UNITS = [("epgx_api.o", "epgx_api.hip", [])] + \
        [(f"epgx_split_p{part}.o", "epgx_split.hip", [f"-DEPGX_PART={part}"]) for part in (16, 0, 8, 4, 2)] + \
        [(f"epgx_cgrow_m{m}.o", "epgx_cgrow.hip", [f"-DEPGX_M={m}"]) for m in (16, 8, 4, 2)] + \
        [(f"epgx_packed_v{v}_k{k}.o", "epgx_packed.hip", [f"-DEPGX_V={v}", f"-DEPGX_KP={k}"]) for v in (3, 2, 1) for k in (32, 16)] + \
        [(f"epgx_deriv_v{v}.o", "epgx_deriv.hip", [f"-DEPGX_V={v}"]) for v in (3, 2, 1)] + \
        [(f"epgx_inst_m{m}.o", "epgx_inst.hip", [f"-DEPGX_M={m}"]) for m in (8, 4, 2, 1, 16)] + \
        [(f"epgx_rows_r{r}.o", "epgx_rows.hip", [f"-DEPGX_R={r}"]) for r in (1, 2, 4, 8)] + \
        [(f"epgx_grow_nsp{n}.o", "epgx_grow.hip", [f"-DEPGX_NSP={n}"]) for n in (1, 2, 4, 0)] + \
        [(f"epgx_rows_deriv_nsp{n}.o", "epgx_rows_deriv.hip", [f"-DEPGX_NSP={n}"]) for n in (0, 1, 2, 4)] + \
        [(f"epgx_rows_deriv_v2_nsp{n}.o", "epgx_rows_deriv.hip", [f"-DEPGX_NSP={n}", "-DEPGX_V=2"]) for n in (0, 1, 2, 4)] + \
        [(f"epgx_drun_v{v}_nsp{n}.o", "epgx_drun.hip", [f"-DEPGX_NSP={n}", f"-DEPGX_V={v}"]) for v in (3, 2, 1) for n in (4, 1)] + \
        [(f"epgx_dfold_v{v}.o", "epgx_dfold.hip", [f"-DEPGX_V={v}"]) for v in (3, 2, 1)] + \
        [(f"epgx_pdfold_v{v}_k{k}.o", "epgx_pdfold.hip", [f"-DEPGX_V={v}", f"-DEPGX_KP={k}"]) for v in (3, 2, 1) for k in (32, 16)]
ARCH = "gfx950"
FLAGS = [f"--offload-arch={ARCH}", "-O3", "-std=c++17", "-fPIC",
         # every branch of run_kernel is wave-uniform (scalar compares on record flags); without this
         # option the AMDGPU backend still structurizes the record dispatch and threads it with mask
         # registers (~20 extra SALU instructions per record)
         "-mllvm", "-structurizecfg-skip-uniform-regions=1",
         # the first 16 dwords of run_kernel's arguments are preloaded into SGPRs at wave launch
         "-mllvm", "-amdgpu-kernarg-preload-count=16"]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libepgx.so cannot be built")
    return exe


STAMP = LIBPATH + ".stamp"     # hash of the sources + flags the library was built from (travels with the library)
_INCLUDE = re.compile(r'^\s*#\s*include\s+"([^"]+)"', re.M)


def include_closure(src):
    """`src` and every file it includes with quotes, transitively (paths relative to csrc/)"""
    seen, todo = [], [src]
    while todo:
        name = os.path.normpath(todo.pop())
        if name in seen:
            continue
        seen.append(name)
        with open(os.path.join(CSRC, name)) as fh:
            text = fh.read()
        todo += [os.path.join(os.path.dirname(name), inc) for inc in _INCLUDE.findall(text)]
    return sorted(seen)


PUBLIC_HEADER = os.path.normpath("../../include/epgx.h")
_COMMENT = re.compile(r"/\*.*?\*/|//[^\n]*", re.S)
_PROTOTYPE = re.compile(r"(?m)^[ \t]*(?:const[ \t]+)?[A-Za-z_][\w \t\*]*\bepgx_\w+[ \t]*\([^;{}]*\)[ \t]*;")


def kernel_view(text):
    """what a KERNEL translation unit sees of include/epgx.h: constants, enums and structs -- comments and the prototypes of
    the C entry points (which only epgx_api.hip defines) removed, white space collapsed.  A new entry point or a reworded
    comment then rebuilds epgx_api.o, not the 45 kernel units (5.5 minutes on 8 cores)"""
    text = _PROTOTYPE.sub("", _COMMENT.sub("", text))
    return " ".join(text.split())


def _dep_bytes(dep, unit_src):
    with open(os.path.join(CSRC, dep), "rb") as fh:
        raw = fh.read()
    if os.path.normpath(dep) == PUBLIC_HEADER and unit_src != "epgx_api.hip":
        return kernel_view(raw.decode()).encode()
    return raw


def depends():
    """every file some translation unit is compiled from (include/epgx.h among them)"""
    return sorted({dep for _, src, _ in UNITS for dep in include_closure(src)})


def unit_hash(unit):
    """sha256 over the translation unit, everything it includes, and its flags: an object is rebuilt only when this changes"""
    import hashlib
    obj, src, extra = unit
    h = hashlib.sha256()
    for dep in include_closure(src):
        h.update(dep.encode())
        h.update(_dep_bytes(dep, src))
    h.update(repr((obj, src, extra, FLAGS)).encode())
    return h.hexdigest()


def source_hash():
    """sha256 over the contents of every source the library depends on, the translation units and the flags"""
    import hashlib
    h = hashlib.sha256()
    for dep in depends():
        h.update(dep.encode())
        with open(os.path.join(CSRC, dep), "rb") as fh:
            h.update(fh.read())
    h.update(repr((UNITS, FLAGS)).encode())
    return h.hexdigest()


def needs_build():
    """True if the library is missing or was built from other sources.  By CONTENT, not by modification time: a snapshot
    copied to another machine or a `git checkout` changes the times of files that did not change"""
    if not os.path.exists(LIBPATH):
        return True
    try:
        with open(STAMP) as fh:
            return fh.read().strip() != source_hash()
    except OSError:
        # a library without a stamp (built by hand): fall back to the modification times
        built = os.path.getmtime(LIBPATH)
        return any(os.path.getmtime(os.path.join(CSRC, d)) > built for d in depends())


def build(force=False, verbose=False, jobs=None):
    """compile the HIP sources into csrc/libepgx.so; returns the library path.  Objects are kept under csrc/build/ with
    the hash of what they were compiled from, so a change to one translation unit recompiles that unit only
    (`force` recompiles everything)"""
    if not force and not needs_build():
        return LIBPATH
    os.makedirs(OBJDIR, exist_ok=True)
    cc = hipcc()

    def compile_unit(unit):
        obj, src, extra = unit
        path = os.path.join(OBJDIR, obj)
        want = unit_hash(unit)
        if not force and os.path.exists(path):
            try:
                with open(path + ".stamp") as fh:
                    if fh.read().strip() == want:
                        return path
            except OSError:
                pass
        cmd = [cc] + FLAGS + extra + ["-c", src, "-o", path]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd, cwd=CSRC)
        with open(path + ".stamp", "w") as fh:
            fh.write(want + "\n")
        return path

    jobs = jobs or min(len(UNITS), max(1, (os.cpu_count() or 2) - 1))
    with concurrent.futures.ThreadPoolExecutor(max_workers=jobs) as pool:
        objects = list(pool.map(compile_unit, UNITS))
    link = [cc, f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", LIBPATH] + objects + ["-ldl"]
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link, cwd=CSRC)
    with open(STAMP, "w") as fh:
        fh.write(source_hash() + "\n")
    return LIBPATH


if __name__ == "__main__":
    import sys
    print(build(force="--force" in sys.argv, verbose=True))
