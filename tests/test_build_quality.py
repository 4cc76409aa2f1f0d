"""Build-quality guard (no GPU): the hot loops of the shipped MFMA kernels must stay free of register-spill traffic.
A scratch reload inside a ring step makes hipcc wait vmcnt(0) there, which drains the LDS-DMA ring every step (it cost
15 % when it happened during development) — and the kernels sit at the 256-VGPR cap, where an innocent epilogue edit can
push a value of the k-loop into scratch.  The check compiles the two kernel files to assembly exactly as csrc/build.sh
does and looks at every basic block that carries MFMAs (tools/hot_spills.py)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "zenker-audio-detection_amd", "csrc")
sys.path.insert(0, os.path.join(ROOT, "tools"))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.mark.skipif(shutil.which(HIPCC) is None and not os.path.exists(HIPCC), reason="hipcc not available")
@pytest.mark.parametrize("src,extra,min_blocks", [("gemm_c8.hip", [], 8), ("attention.hip", ["-fno-honor-nans"], 3)])
def test_mfma_blocks_carry_no_scratch_traffic(tmp_path, src, extra, min_blocks):
    from hot_spills import loop_scratch, mfma_blocks
    out = tmp_path / (src + ".s")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function", "-mllvm", "-amdgpu-mfma-vgpr-form",
           "-I" + CSRC, *extra, "-S", "--cuda-device-only", os.path.join(CSRC, src), "-o", str(out)]
    subprocess.run(cmd, check=True, capture_output=True, timeout=900)
    text = out.read_text()
    # (1) attention: nothing inside any loop touches scratch (the key-tile loops are the only loops there are)
    loops = loop_scratch(text)
    assert loops, "no kernels found in the assembly"
    for kern, (nb, nm, sl, ss, outside) in loops.items():
        if src == "gemm_c8.hip":      # (persistent tile loop: its epilogue lies inside the outer loop; rule (2) guards the ring steps)
            continue
        assert sl == 0 and ss == 0, f"{kern}: {sl} scratch loads / {ss} scratch stores inside a loop ({nm} MFMAs in {nb} loop blocks)"
        # outside the loops (prologue, the tail tiles behind the tile loop, epilogue) a few spilled values per workgroup are tolerated
        assert outside <= 16, f"{kern}: {outside} scratch instructions outside its loops"
    assert sum(nm for (_nb, nm, _sl, _ss, _o) in loops.values()) >= 40
    # (2) the long MFMA-carrying blocks, wherever they lie (the GEMM's ring steps; attention's score halves)
    res = mfma_blocks(text, 20 if src == "gemm_c8.hip" else 12)
    n = 0
    for kern, rows in res.items():
        for (label, mfma, sld, sst, dma, vm) in rows:
            n += 1
            if src == "gemm_c8.hip":
                assert sld == 0 and sst == 0, f"{kern} block {label}: {sld} scratch loads / {sst} scratch stores beside {mfma} MFMAs"
    assert n >= min_blocks, (n, res)
