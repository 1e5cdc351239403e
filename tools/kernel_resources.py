#!/usr/bin/env python3
"""Per-kernel register / LDS / scratch usage of the built libmoeinf_hip.so (no GPU needed): pulls the gfx950 code
object out of the fat binary and reads its metadata notes.  usage: tools/kernel_resources.py [substring ...]"""
import os, re, struct, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
data = open(os.path.join(ROOT, "moe-infinity_amd", "libmoeinf_hip.so"), "rb").read()
cos = []
i = data.find(b"__CLANG_OFFLOAD_BUNDLE__")
while i >= 0:  # one bundle per translation unit
    n = struct.unpack_from("<Q", data, i + 24)[0]
    off = i + 32
    for _ in range(n):
        o, s, l = struct.unpack_from("<QQQ", data, off); off += 24
        name = data[off:off + l].decode(); off += l
        if "gfx950" in name:
            cos.append(data[i + o:i + o + s])
    i = data.find(b"__CLANG_OFFLOAD_BUNDLE__", i + 24)
demangle = lambda m: subprocess.run(["c++filt", m], capture_output=True, text=True).stdout.strip()
for co in cos:
    with tempfile.NamedTemporaryFile(suffix=".co") as f:
        f.write(co); f.flush()
        notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], capture_output=True, text=True).stdout
    for k in re.split(r"\n\s+- \.agpr_count", notes)[1:]:
        name = re.search(r"\.name:\s+(\S+)", k).group(1)
        g = lambda fld: int(re.search(r"\." + fld + r":\s+(\d+)", k).group(1))
        d = demangle(name).split("(")[0].replace("void moeinf::", "").replace("unsigned short", "bf16")
        if sys.argv[1:] and not any(a in d for a in sys.argv[1:]):
            continue
        print(f"{d:80s} vgpr {g('vgpr_count'):3d} sgpr {g('sgpr_count'):3d} lds {g('group_segment_fixed_size'):6d} scratch {g('private_segment_fixed_size')}")
