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
    from hot_spills import mfma_blocks
    out = tmp_path / (src + ".s")
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-function", "-mllvm", "-amdgpu-mfma-vgpr-form",
           "-I" + CSRC, *extra, "-S", "--cuda-device-only", os.path.join(CSRC, src), "-o", str(out)]
    subprocess.run(cmd, check=True, capture_output=True, timeout=900)
    res = mfma_blocks(out.read_text(), 20 if src == "gemm_c8.hip" else 12)
    assert res, "no kernels found in the assembly"
    n = 0
    for kern, rows in res.items():
        for (label, mfma, sld, sst, dma, vm) in rows:
            n += 1
            assert sld == 0 and sst == 0, f"{kern} block {label}: {sld} scratch loads / {sst} scratch stores beside {mfma} MFMAs"
    assert n >= min_blocks, (n, res)
