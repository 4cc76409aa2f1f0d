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


if __name__ == "__main__":
    res = mfma_blocks(open(sys.argv[1]).read(), int(sys.argv[2]) if len(sys.argv) > 2 else 8)
    for k, rows in res.items():
        print(k[:60], "\n   (block, mfma, scratch_load, scratch_store, lds_dma, vmcnt waits):", rows)
