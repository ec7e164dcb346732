#!/usr/bin/env python3
"""ISA of one kernel of libxmca_hip.so's gfx950 code object (no GPU needed):  disasm_kernel.py <substring of the mangled name> [out.s]"""
import os, struct, subprocess, sys, tempfile
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
data = open(os.path.join(REPO, "xmca_amd", "libxmca_hip.so"), "rb").read()
i = data.find(b"__CLANG_OFFLOAD_BUNDLE__")
off = i + 24
n = struct.unpack_from("<Q", data, off)[0]; off += 8
co = None
for _ in range(n):
    o, sz, tl = struct.unpack_from("<QQQ", data, off); off += 24
    t = data[off:off + tl].decode(); off += tl
    if "gfx950" in t:
        co = data[i + o:i + o + sz]
f = tempfile.NamedTemporaryFile(suffix=".co", delete=False); f.write(co); f.close()
OD = "/opt/rocm/lib/llvm/bin/llvm-objdump"
syms = subprocess.run([OD, "-t", f.name], capture_output=True, text=True).stdout
names = [l.split()[-1] for l in syms.splitlines() if " F .text" in l and sys.argv[1] in l]
for nm in names:
    s = subprocess.run([OD, "-d", "--disassemble-symbols=" + nm, f.name], capture_output=True, text=True).stdout
    out = sys.argv[2] if len(sys.argv) > 2 else None
    if out: open(out, "a").write(s)
    else: print(s)
    print(nm, len(s.splitlines()), "lines", file=sys.stderr)
os.unlink(f.name)
