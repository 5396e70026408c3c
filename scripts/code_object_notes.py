#!/usr/bin/env python3
"""Per-kernel resource notes of the gfx950 code objects inside lib/libtmpc_hip.so (no GPU needed).

The shared library carries one clang offload bundle per translation unit in its .hip_fatbin section; every bundle holds
an AMDGPU ELF whose NT_AMDGPU_METADATA note (msgpack) lists, per kernel, the registers, spills, the private (scratch)
segment and the LDS the compiler settled on.  `kernels(path)` returns {demangled-ish name: note dict}."""
import struct
import subprocess
import sys

import msgpack

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def _section(path, name):
    """bytes of an ELF64 section of the host library"""
    data = open(path, "rb").read()
    shoff, = struct.unpack_from("<Q", data, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", data, 0x3A)
    def sh(i):
        return struct.unpack_from("<IIQQQQIIQQ", data, shoff + i * shentsize)
    stroff = sh(shstrndx)[4]
    for i in range(shnum):
        nm, _, _, _, off, size = sh(i)[:6]
        end = data.index(b"\0", stroff + nm)
        if data[stroff + nm:end].decode() == name:
            return data[off:off + size]
    raise KeyError(name)


def code_objects(path):
    fat = _section(path, ".hip_fatbin")
    out = []
    pos = fat.find(MAGIC)
    while pos >= 0:
        n, = struct.unpack_from("<Q", fat, pos + len(MAGIC))
        p = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, idlen = struct.unpack_from("<QQQ", fat, p)
            ident = fat[p + 24:p + 24 + idlen].decode()
            p += 24 + idlen
            if "amdgcn" in ident and size:
                out.append((ident, fat[pos + off:pos + off + size]))
        pos = fat.find(MAGIC, pos + 1)
    return out


def _notes(elf):
    shoff, = struct.unpack_from("<Q", elf, 0x28)
    shentsize, shnum, _ = struct.unpack_from("<HHH", elf, 0x3A)
    for i in range(shnum):
        _, typ, _, _, off, size = struct.unpack_from("<IIQQQQ", elf, shoff + i * shentsize)
        if typ != 7:      # SHT_NOTE
            continue
        p = off
        while p < off + size:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
            p += 12
            name = elf[p:p + namesz].rstrip(b"\0")
            p += (namesz + 3) // 4 * 4
            desc = elf[p:p + descsz]
            p += (descsz + 3) // 4 * 4
            if name == b"AMDGPU" and ntype == 32:      # NT_AMDGPU_METADATA
                yield msgpack.unpackb(desc, raw=False, strict_map_key=False)


def kernels(path):
    res = {}
    for _, elf in code_objects(path):
        for meta in _notes(elf):
            for k in meta.get("amdhsa.kernels", []):
                res[k[".name"]] = k
    return res


def demangle(names):
    try:
        out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True).stdout.split("\n")
        return dict(zip(names, out))
    except Exception:
        return {n: n for n in names}


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else "robust-tracking-mpc-over-lossy-networks_amd/lib/libtmpc_hip.so"
    ks = kernels(lib)
    dm = demangle(list(ks))
    print("%-70s %5s %5s %6s %8s %8s" % ("kernel", "vgpr", "agpr", "spill", "scratch", "lds"))
    for n, k in sorted(ks.items(), key=lambda kv: dm[kv[0]]):
        short = dm[n].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").replace("tmpc::", "")
        print("%-70s %5d %5d %6d %8d %8d" % (short[:70], k[".vgpr_count"], k.get(".agpr_count", 0), k[".vgpr_spill_count"],
                                               k[".private_segment_fixed_size"], k[".group_segment_fixed_size"]))
