"""Probe libraries (libzkast_probes*.so) are linked like the product: without a DT_NEEDED on libamdhip64.  Load the ONE HIP
runtime of the process first (zkast/lib.py::_ensure_hip_runtime), then the probe library."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "zenker-audio-detection_amd"))


def cdll(path):
    from zkast import lib
    lib._ensure_hip_runtime()
    return C.CDLL(path)
