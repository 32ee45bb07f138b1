"""Print the instruction-class shape of a kernel's MFMA loop from a hipcc -save-temps .s file.

    python tools/isa_shape.py conv_bf3-hip-amdgcn-amd-amdhsa-gfx950.s p16_kernelILi128ELi128ELb0 [min_mfma]

M = MFMA, r/w = ds_read/ds_write, G = buffer load to VGPR, L = buffer load to LDS (DMA), v = VALU,
s = SALU, B = barrier, |..| = s_waitcnt, J = branch.  One line per basic block with >= min_mfma MFMAs.
"""
import sys


def main():
    s = open(sys.argv[1]).read()
    key = sys.argv[2]
    names = [ln.split(":")[0] for ln in s.split("\n") if key in ln and ln.startswith("_Z") and ":" in ln]
    i = s.index("\n" + names[0] + ":")
    j = s.index(".end_amdhsa_kernel", i)
    out, cur = [], []
    for ln in s[i:j].split("\n"):
        t = ln.strip()
        if not t or t.startswith(";"):
            continue
        t = t.split(";")[0].strip()
        if not t:
            continue
        if t.endswith(":") or t.startswith(".LBB"):
            out.append("".join(cur)); cur = []
            continue
        if t.startswith("."):
            continue
        op = t.split()[0]
        if op.startswith("v_mfma"): c = "M"
        elif op.startswith("ds_read"): c = "r"
        elif op.startswith("ds_write"): c = "w"
        elif op.startswith("buffer_load"): c = "L" if " lds" in t else "G"
        elif op.startswith("s_waitcnt"): c = "|" + t.split(None, 1)[1].replace(" ", "") + "|"
        elif op.startswith("s_barrier"): c = "B"
        elif op.startswith("v_"): c = "v"
        elif "branch" in op: c = "J"
        elif op.startswith("s_"): c = "s"
        else: c = "?"
        cur.append(c)
    out.append("".join(cur))
    mn = int(sys.argv[3]) if len(sys.argv) > 3 else 32
    for blk in out:
        if blk.count("M") >= mn:
            print(blk, "\n")


if __name__ == "__main__":
    main()
