"""List, per kernel of a gfx950 .s file, the basic blocks that hold MFMAs together with their scratch (spill) traffic.
usage: python tools/hot_spills.py file.s [min_mfma=8]"""
import re
import sys

s = open(sys.argv[1]).read()
min_mfma = int(sys.argv[2]) if len(sys.argv) > 2 else 8
for m in re.finditer(r'^(_Z\w+):[^\n]*\n(.*?)s_endpgm', s, re.S | re.M):
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
    print(m.group(1)[:60], "\n   (block, mfma, scratch_load, scratch_store, lds_dma, vmcnt waits):", rows)
