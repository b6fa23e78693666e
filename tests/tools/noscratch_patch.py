"""Developer experiment: a copy of the library in which one kernel's private segment (scratch) is declared empty.

The register allocator leaves a DEAD 64-byte spill slot in the S64 plan kernel's frame (a slot it made for a 16-register tuple and then
did not use: no instruction of the kernel touches scratch -- tests/test_kernel_resources.py checks that); the kernel descriptor and the
code object's metadata still say 68 bytes of private segment per lane, so every dispatch sets scratch up.  This tool zeroes both
(descriptor: private_segment_fixed_size and ENABLE_PRIVATE_SEGMENT of compute_pgm_rsrc2; metadata: .private_segment_fixed_size) in a COPY
of the library, to measure what the set-up costs (tests/tools/ab.py).  It refuses when the kernel's code holds a scratch instruction.
    python tests/tools/noscratch_patch.py in.so out.so <substring of the mangled kernel name>"""
import struct
import subprocess
import sys


def sections(d, base):
    shoff, = struct.unpack_from("<Q", d, base + 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", d, base + 0x3A)
    secs = []
    for i in range(shnum):
        name, typ, flags, addr, off, size, link, info, align, entsize = struct.unpack_from("<IIQQQQIIQQ", d, base + shoff + i * shentsize)
        secs.append(dict(name=name, type=typ, addr=addr, off=off, size=size, link=link, entsize=entsize))
    stro = secs[shstrndx]["off"]
    for s in secs:
        e = d.index(b"\0", base + stro + s["name"])
        s["name"] = d[base + stro + s["name"]:e].decode()
    return secs


def main(src, dst, want):
    d = bytearray(open(src, "rb").read())
    host = sections(d, 0)
    fat = next(s for s in host if s["name"] == ".hip_fatbin")
    assert d[fat["off"]:fat["off"] + 24] == b"__CLANG_OFFLOAD_BUNDLE__"
    n, = struct.unpack_from("<Q", d, fat["off"] + 24)
    o = fat["off"] + 32
    co = None
    for _ in range(n):
        off, size, tl = struct.unpack_from("<QQQ", d, o)
        o += 24
        triple = bytes(d[o:o + tl])
        o += tl
        if b"gfx950" in triple:
            co = fat["off"] + off
    assert co is not None and d[co:co + 4] == b"\x7fELF"
    secs = sections(d, co)
    symtab = next(s for s in secs if s["name"] == ".symtab")
    strtab = secs[symtab["link"]]
    kd = None
    for i in range(symtab["size"] // 24):
        nm, info, other, shndx, value, size = struct.unpack_from("<IBBHQQ", d, co + symtab["off"] + 24 * i)
        e = d.index(b"\0", co + strtab["off"] + nm)
        name = d[co + strtab["off"] + nm:e].decode()
        if name.endswith(".kd") and want in name:
            assert kd is None, "the substring names several kernels"
            kd = (name[:-3], value, shndx)
    assert kd, "no such kernel"
    name, value, shndx = kd
    # no scratch instruction in the kernel's code
    dis = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "--disassemble-symbols=" + name, "/dev/stdin"], input=bytes(d[co:]),
                         capture_output=True).stdout.decode()
    assert dis.count("\n") > 1000, "disassembly failed"
    bad = [l for l in dis.split("\n") if "scratch_" in l or ("buffer_" in l and "buffer_inv" not in l and "buffer_wbl2" not in l) or "flat_scratch" in l]
    assert not bad, bad[:5]
    sec = secs[shndx]
    at = co + sec["off"] + (value - sec["addr"])
    priv, = struct.unpack_from("<I", d, at + 4)
    rsrc2, = struct.unpack_from("<I", d, at + 0x34)
    print(name[:80], "descriptor: private segment", priv, "rsrc2 %#x" % rsrc2)
    struct.pack_into("<I", d, at + 4, 0)
    struct.pack_into("<I", d, at + 0x34, rsrc2 & ~1)
    # metadata (msgpack): ... ".name" <str> ".private_segment_fixed_size" <uint> ... of the same map
    key = b".name"
    needle = name.encode()
    note = next(s for s in secs if s["name"] == ".note")
    lo, hi = co + note["off"], co + note["off"] + note["size"]
    p = d.index(needle, lo, hi)
    while d[p - 8:p].find(key) < 0:  # (the symbol entry ".symbol" holds name + ".kd"; the ".name" entry is the one right behind its key)
        p = d.index(needle, p + 1, hi)
    q = d.index(b".private_segment_fixed_size", p, hi) + len(b".private_segment_fixed_size")
    assert d[q] == priv and priv < 128, (d[q], priv)
    d[q] = 0
    open(dst, "wb").write(d)
    print("written", dst)


if __name__ == "__main__":
    main(*sys.argv[1:4])
