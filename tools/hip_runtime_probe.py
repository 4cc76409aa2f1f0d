"""Which HIP / HSA runtime images end up in ONE process that uses both libzkast.so and torch — and does the GPU come up?

Round 3: tests/test_model_gpu.py on its own ended in torch's "RuntimeError: No HIP GPUs are available" after libzkast.so had
already run on the GPU; conftest.py hid it by bringing torch's device context up first.  This probe runs every order in a
FRESH process and prints /proc/self/maps' libamdhip64 / libhsa-runtime64 images:
   old link  = libzkast.so with DT_NEEDED libamdhip64.so.7 + RUNPATH /opt/rocm (rounds 1-3; tools builds it as
               zkast/libzkast_oldlink.so: same objects, linked without -no-hip-rt, loaded WITHOUT _ensure_hip_runtime)
   new link  = the product (csrc/build.sh: -no-hip-rt; zkast/lib.py::_ensure_hip_runtime picks the one runtime)
usage (GPU box): python tools/hip_runtime_probe.py > gpurun_out/hip_runtime_probe.txt"""
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OLD = os.path.join(ROOT, "zenker-audio-detection_amd", "zkast", "libzkast_oldlink.so")

COMMON = """
import os, sys, ctypes as C
sys.path.insert(0, os.path.join(%r, "zenker-audio-detection_amd"))
import numpy as np
def images():
    out = []
    for line in open("/proc/self/maps"):
        p = line.rstrip("\\n").split(None, 5)[-1] if "/" in line else ""
        b = os.path.basename(p)
        if (b.startswith("libamdhip64") or b.startswith("libhsa-runtime64") or b.startswith("librccl")) and p not in out:
            out.append(p)
    return out
def zk_work():
    from zkast import lib, synth
    ctx = lib.get_context(0)
    rec = synth.synth_recording(1, 16000 * 3)
    ctx.logmel(rec, rec.size, 0, 8000, 16000, 5)
    f = ctx.features_get()
    return float(np.abs(f).sum())
def torch_work():
    import torch
    return float((torch.ones(1000, device="cuda") * 2).sum().item())
""" % ROOT

OLD_PATCH = """
from zkast import lib as _l
_l._ensure_hip_runtime = lambda: "(old link: DT_NEEDED libamdhip64.so.7, RUNPATH /opt/rocm)"
_l.LIB_PATH = %r
""" % OLD

CASES = [
    ("old link, zkast first, then torch", OLD_PATCH + "print('zk', zk_work()); print('images', images()); print('torch', torch_work())"),
    ("old link, torch first, then zkast", OLD_PATCH + "print('torch', torch_work()); print('zk', zk_work())"),
    ("new link, zkast first, then torch", "print('zk', zk_work()); print('images', images()); print('torch', torch_work())"),
    ("new link, torch first, then zkast", "print('torch', torch_work()); print('zk', zk_work())"),
    ("new link, zkast + RCCL world of one, then torch",
     "from zkast import lib\nctx = lib.get_context(0)\nctx.comm_init(0, 1, None)\nuid = lib.comm_unique_id()\nprint('rccl id', len(uid))\n"
     "print('zk', zk_work()); print('torch', torch_work())"),
]


def main():
    for name, body in CASES:
        if "old link" in name and not os.path.exists(OLD):
            print(f"== {name}: skipped ({OLD} not built)")
            continue
        code = COMMON + textwrap.dedent(body) + "\nprint('images at exit', images())\n"
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
        print(f"== {name}: exit {r.returncode}")
        for ln in r.stdout.splitlines():
            print("   " + ln)
        if r.returncode:
            for ln in r.stderr.strip().splitlines()[-4:]:
                print("   ! " + ln)
        sys.stdout.flush()


if __name__ == "__main__":
    main()
