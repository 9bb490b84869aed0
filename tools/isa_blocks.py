"""Per-basic-block instruction statistics of one kernel's ISA (hipcc -S --cuda-device-only output):
    python tools/isa_blocks.py file.s mangled-kernel-name-substring
MFMAs, SGPR-spill traffic (v_readlane / v_writelane), scratch accesses, scalar / vector ALU, LDS and global
instructions per block - where between the matrix instructions the scalar pipe and the spill code sit."""
import re
import sys


def kernel_text(path, sub):
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*:", l) and sub in l)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return lines[start:end]


def blocks(lines):
    segs, cur = [], ("entry", 0, [])
    for i, l in enumerate(lines):
        m = re.match(r"^(\.LBB\d+_\d+):(.*)", l)
        if m:
            segs.append(cur)
            cur = (m.group(1) + " " + m.group(2).strip(), i, [])
        cur[2].append(l)
    segs.append(cur)
    return segs


PAT = dict(mfma=r"\bv_mfma", readlane=r"\bv_readlane", writelane=r"\bv_writelane", scratch=r"\bscratch_",
           s_load=r"\bs_load", salu=r"\bs_(?!waitcnt|nop|barrier|load|cbranch|branch|endpgm|setprio|sleep)",
           valu=r"\bv_(?!mfma|readlane|writelane|accvgpr)", acc=r"\bv_accvgpr", ds=r"\bds_", glob=r"\bglobal_",
           wait=r"\bs_waitcnt", nop=r"\bs_nop", barrier=r"\bs_barrier")

if __name__ == "__main__":
    ls = kernel_text(sys.argv[1], sys.argv[2])
    tot = {k: 0 for k in PAT}
    for name, start, body in blocks(ls):
        t = "\n".join(body)
        c = {k: len(re.findall(p, t)) for k, p in PAT.items()}
        for k in tot:
            tot[k] += c[k]
        if c["mfma"] or c["readlane"] > 3 or c["writelane"] > 3 or c["scratch"]:
            print(f"{name[:40]:40s} @{start:6d} n={len(body):5d} " + " ".join(f"{k}={v}" for k, v in c.items() if v))
    print("TOTAL", tot)
