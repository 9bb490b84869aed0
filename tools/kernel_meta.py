"""Per-kernel resources of a built library, read from the code objects inside it (no GPU, no ROCm tool):
    python tools/kernel_meta.py [library] [substring ...]
name, VGPRs, AGPRs, scratch bytes, LDS bytes, max workgroup size - what the register allocator actually did."""
import struct
import sys

MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(blob):
    """gfx950 ELF images of every offload bundle in ``blob``"""
    out, pos = [], 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            return out
        n = struct.unpack_from("<Q", blob, pos + len(MAGIC))[0]
        cur = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, cur)
            triple = blob[cur + 24: cur + 24 + tlen].decode()
            cur += 24 + tlen
            if "gfx950" in triple and size > 0:
                out.append(blob[pos + off: pos + off + size])
        pos = cur


def kernels(elf):
    """[{name, vgpr, agpr, sgpr, scratch, lds, max_wg}] from the NT_AMDGPU_METADATA note (msgpack)"""
    import msgpack

    assert elf[:4] == b"\x7fELF"
    shoff = struct.unpack_from("<Q", elf, 0x28)[0]
    shentsize, shnum = struct.unpack_from("<HH", elf, 0x3A)
    res = []
    for i in range(shnum):
        sh = shoff + i * shentsize
        sh_type = struct.unpack_from("<I", elf, sh + 4)[0]
        off, size = struct.unpack_from("<QQ", elf, sh + 0x18)
        if sh_type != 7:  # SHT_NOTE
            continue
        p = off
        while p + 12 <= off + size:
            namesz, descsz, ntype = struct.unpack_from("<III", elf, p)
            p += 12
            name = elf[p: p + namesz]
            p += (namesz + 3) & ~3
            desc = elf[p: p + descsz]
            p += (descsz + 3) & ~3
            if ntype == 32 and name.startswith(b"AMDGPU"):
                md = msgpack.unpackb(desc, raw=False, strict_map_key=False)
                for k in md.get("amdhsa.kernels", []):
                    res.append(dict(name=k[".name"], vgpr=k.get(".vgpr_count", 0), agpr=k.get(".agpr_count", 0),
                                    sgpr=k.get(".sgpr_count", 0), scratch=k.get(".private_segment_fixed_size", 0),
                                    lds=k.get(".group_segment_fixed_size", 0),
                                    sgpr_spill=k.get(".sgpr_spill_count", 0), vgpr_spill=k.get(".vgpr_spill_count", 0),
                                    max_wg=k.get(".max_flat_workgroup_size", 0)))
    return res


def library_kernels(path):
    blob = open(path, "rb").read()
    out = []
    for co in code_objects(blob):
        out.extend(kernels(co))
    return out


def demangle(names):
    import subprocess

    for tool in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "c++filt"):
        try:
            r = subprocess.run([tool], input="\n".join(names), capture_output=True, text=True)
            d = r.stdout.splitlines()
            if r.returncode == 0 and len(d) == len(names):
                return d
        except Exception:
            pass
    return names


if __name__ == "__main__":
    import os

    lib = sys.argv[1] if len(sys.argv) > 1 and os.path.exists(sys.argv[1]) else os.path.join(
        os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "spatial_alignment_amd", "libgpsa_hip.so")
    pats = [a for a in sys.argv[1:] if not os.path.exists(a)]
    ks = library_kernels(lib)
    for k, nm in zip(ks, demangle([k["name"] for k in ks])):
        if pats and not any(p in nm for p in pats):
            continue
        print(f"{nm[:110]:110s} vgpr {k['vgpr']:4d} agpr {k['agpr']:4d} scratch {k['scratch']:6d} lds {k['lds']:6d} wg {k['max_wg']}")
