"""Static resources of every kernel in a libbndm_hip.so, from the code objects' metadata and kernel descriptors (no GPU): registers, LDS,
scratch, and the occupancy they allow on gfx950 (512 unified VGPR+AGPR per lane and SIMD, allocated in blocks of 8; 160 KiB LDS per CU;
at most 8 waves per SIMD).  Dynamic LDS is a launch parameter: taken from recorded launch traces when given (first trace first).
    python tools/kernel_resources.py [lib.so] [trace.txt ...]  >  profiles/rNN_kernel_resources.txt"""
import hashlib
import os
import re
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    from tests.gfx950sim import loader
    args = sys.argv[1:]
    lib = args[0] if args and args[0].endswith(".so") else "bndm_amd/libbndm_hip.so"
    traces = [a for a in args if not a.endswith(".so")]
    ks = loader.load_library(lib)
    dyn, block = {}, {}
    for t in traces:
        seen_before = set(dyn)
        for ln in open(t):
            if ln.startswith("launch "):
                m = re.match(r"launch (\S+) g=\S+ b=(\d+),(\d+),(\d+) lds=(\d+)", ln)
                if m and m.group(1) not in seen_before:          # the FIRST trace that launches a kernel decides (give the benchmark's first)
                    dyn[m.group(1)] = max(dyn.get(m.group(1), 0), int(m.group(5)))
                    block[m.group(1)] = int(m.group(2)) * int(m.group(3)) * int(m.group(4))
    # the 4-wave conv_t32 variants are picked by the library at >= 448 workgroups (batch 64): block / LDS size as the library launches them
    from tests.gfx950sim import suite
    for ent in (suite.S4 + ";" + suite.S4B).split(";"):
        to, b_, l_ = ent.split("=>")[1].split(":")
        for k in ks:
            if to in k:
                dyn.setdefault(k, int(l_))
                block.setdefault(k, int(b_))
    short = lambda k: re.sub(r"^_ZN4bndm12_GLOBAL__N_1\d+|^_ZN12_GLOBAL__N_1\d+", "", k)
    print(f"# {lib}  sha256 {hashlib.sha256(open(lib, 'rb').read()).hexdigest()}")
    print("# arch VGPRs / AGPRs / SGPRs from the metadata notes; unified = the allocation of one wave (VGPRs rounded up to 4 + AGPRs, in blocks of 8);")
    print("# waves/SIMD by registers = min(8, 512 // unified); LDS = static + the dynamic size of the first given trace that launches the kernel (its largest there; - = never launched);")
    print("# WG/CU by LDS = 160 KiB // LDS; waves/SIMD = what both limits and the block size allow (4 SIMDs per CU)")
    print(f"# {'kernel':70s} {'vgpr':>4s} {'agpr':>4s} {'sgpr':>4s} {'unified':>7s} {'w/SIMD(reg)':>11s} {'scratch':>7s} {'LDS B':>7s} {'block':>5s} {'WG/CU(LDS)':>10s} {'w/SIMD':>6s} {'insts':>6s}")
    for k in sorted(ks, key=short):
        ki = ks[k]
        uni = -(-((-(-ki.vgpr_count // 4) * 4) + ki.agpr_count) // 8) * 8 if ki.agpr_count else -(-ki.vgpr_count // 8) * 8
        wreg = min(8, 512 // max(uni, 8))
        d = dyn.get(k)
        lds = ki.lds_static + (d or 0)
        b = block.get(k)
        wg_lds = (160 * 1024) // lds if lds else 99
        occ = "-"
        if b:
            wpw = -(-b // 64)                                 # waves per workgroup
            wg_reg = (wreg * 4) // wpw
            occ = f"{min(wg_reg, wg_lds, 32) * wpw / 4:.1f}"
        print(f"{short(k)[:72]:72s} {ki.vgpr_count:4d} {ki.agpr_count:4d} {ki.sgpr_count:4d} {uni:7d} {wreg:11d} {ki.scratch:7d} "
              f"{(str(lds) if d is not None or ki.lds_static else '-'):>7s} {(str(b) if b else '-'):>5s} {(str(wg_lds) if lds else '-'):>10s} {occ:>6s} {len(ki.insts):6d}")


if __name__ == "__main__":
    main()
