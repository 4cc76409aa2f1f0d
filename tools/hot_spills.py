"""List, per kernel of a gfx950 .s file, the basic blocks that hold MFMAs together with their scratch (spill) traffic.
A spill reload inside a ring step makes hipcc wait vmcnt(0) there and drains the LDS-DMA ring (-15 % when it happened),
so tests/test_build_quality.py asserts that the MFMA blocks of the shipped kernels are free of scratch traffic.
usage: python tools/hot_spills.py file.s [min_mfma=8]"""
import re
import sys


def mfma_blocks(asm_text, min_mfma=8):
    """{kernel symbol: [(block label, mfma, scratch_load, scratch_store, lds_dma, vmcnt waits), ...]}"""
    out = {}
    for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)s_endpgm', asm_text, re.S | re.M):
        blocks, cur = [], ["entry"]
        for l in m.group(2).split('\n'):
            if re.match(r'^\.LBB', l):
                blocks.append(cur); cur = []
            cur.append(l)
            if re.search(r's_cbranch|s_branch|s_setpc', l):      # a branch ends the block as well
                blocks.append(cur); cur = [cur[0] + "+"]
        blocks.append(cur)
        rows = []
        for b in blocks:
            nm = sum('v_mfma' in l for l in b)
            if nm >= min_mfma:
                rows.append((b[0].split(':')[0], nm, sum('scratch_load' in l for l in b), sum('scratch_store' in l for l in b),
                             sum('global_load_lds' in l for l in b), sum('s_waitcnt vmcnt' in l for l in b)))
        out[m.group(1)] = rows
    return out


def loop_scratch(asm_text):
    """{kernel symbol: (blocks inside loops, MFMAs inside loops, scratch loads inside loops, scratch stores inside loops, scratch ops
    outside loops)} — LLVM marks every block of a loop body with "in Loop:" (the header with "Loop Header").  Since round 5
    attention's PV slots hold scalar branches (the skipped Vl·P MFMAs), so its MFMA-carrying blocks are short and the
    per-block rule above no longer sees them all; what matters is this: NO scratch traffic anywhere inside a loop."""
    out = {}
    for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)s_endpgm', asm_text, re.S | re.M):
        in_loop = False
        nb = nm = sl = ss = outside = 0
        for l in m.group(2).split('\n'):
            if re.match(r'^\.LBB', l):
                in_loop = ("in Loop" in l) or ("Loop Header" in l)
                nb += in_loop
            if in_loop:
                nm += 'v_mfma' in l; sl += 'scratch_load' in l; ss += 'scratch_store' in l
            else:
                outside += ('scratch_load' in l) or ('scratch_store' in l)
        out[m.group(1)] = (nb, nm, sl, ss, outside)
    return out


if __name__ == "__main__":
    res = mfma_blocks(open(sys.argv[1]).read(), int(sys.argv[2]) if len(sys.argv) > 2 else 8)
    for k, rows in res.items():
        print(k[:60], "\n   (block, mfma, scratch_load, scratch_store, lds_dma, vmcnt waits):", rows)
    for k, v in loop_scratch(open(sys.argv[1]).read()).items():
        print(k[:60], "\n   (loop blocks, MFMAs in loops, scratch loads / stores in loops, scratch ops outside loops):", v)
