#!/usr/bin/env python3
"""Registers / scratch of every kernel in the gfx950 code object of libxmca_hip.so (no GPU needed).
usage: code_object_report.py [substring ...]   -> name, vgpr, agpr, sgpr, spilled vgprs, scratch bytes, LDS bytes"""
import os, re, struct, subprocess, sys, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = os.path.join(REPO, "xmca_amd", "libxmca_hip.so")
data = open(lib, "rb").read()
i = data.find(b"__CLANG_OFFLOAD_BUNDLE__")
off = i + 24
n = struct.unpack_from("<Q", data, off)[0]; off += 8
co = None
for _ in range(n):
    o, sz, tl = struct.unpack_from("<QQQ", data, off); off += 24
    t = data[off:off + tl].decode(); off += tl
    if "gfx950" in t:
        co = data[i + o:i + o + sz]
assert co is not None, "no gfx950 code object"
with tempfile.NamedTemporaryFile(suffix=".co", delete=False) as f:
    f.write(co)
notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", f.name], capture_output=True, text=True).stdout
os.unlink(f.name)
demangle = lambda s: subprocess.run(["c++filt", s], capture_output=True, text=True).stdout.strip()
want = sys.argv[1:]
rows = []
for k in re.split(r"\n\s+- \.agpr_count", notes)[1:]:
    g = lambda key: (re.search(r"\.%s:\s+(\S+)" % key, k) or [None, "?"])[1]
    name = g("name")
    if want and not any(w in name for w in want):
        continue
    rows.append((demangle(name)[:110], g("vgpr_count"), g("sgpr_count"), g("vgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
print("%-110s %5s %5s %6s %8s %7s" % ("kernel", "vgpr", "sgpr", "spill", "scratch", "lds"))
for r in sorted(rows):
    print("%-110s %5s %5s %6s %8s %7s" % r)
print("kernels with spills:", sum(1 for r in rows if r[3] not in ("0", "?")), "of", len(rows))
