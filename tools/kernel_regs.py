#!/usr/bin/env python3
"""VGPRs / scratch bytes / spills of the kernels inside libepgx.so (reads the gfx950 code objects out of the offload
bundles and their AMDGPU metadata notes):   python tools/kernel_regs.py [substring ...]"""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.environ.get("EPGX_LIBRARY", os.path.join(ROOT, "epgpy_amd", "csrc", "libepgx.so"))
READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"


def main():
    want = sys.argv[1:]
    data = open(LIB, "rb").read()
    rows = []
    for m in re.finditer(b"__CLANG_OFFLOAD_BUNDLE__", data):
        i = m.start()
        n = struct.unpack_from("<Q", data, i + 24)[0]
        p = i + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", data, p)
            p += 24
            triple = data[p:p + tl].decode()
            p += tl
            if "gfx950" not in triple or not size:
                continue
            with tempfile.NamedTemporaryFile(suffix=".co") as f:
                f.write(data[i + off:i + off + size])
                f.flush()
                txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
            cur = {}
            for line in txt.splitlines():
                mm = re.match(r"\s*-?\s*\.(name|vgpr_count|private_segment_fixed_size|sgpr_count|vgpr_spill_count|max_flat_workgroup_size):\s*(.*)", line)
                if mm:
                    cur[mm.group(1)] = mm.group(2).strip()
                if ".wavefront_size" in line and cur:
                    rows.append(cur)
                    cur = {}
    names = subprocess.run(["c++filt"], input="\n".join(r.get("name", "") for r in rows), capture_output=True, text=True).stdout.splitlines()
    for r, name in zip(rows, names):
        name = name.replace("epgx::", "").replace("void ", "")
        name = re.sub(r"\(.*$", "", name)
        if want and not any(w in name for w in want):
            continue
        print(f"{name:70s} vgpr {r.get('vgpr_count', '?'):>4s}  scratch {r.get('private_segment_fixed_size', '?'):>5s}  spills {r.get('vgpr_spill_count', '0'):>3s}")


if __name__ == "__main__":
    main()
