"""zkast — MI355X-native (gfx950) two-stage AST audio-window inference.

Host-side mirror of the hot path of daostler-tum/zenker-audio-detection
(src/test_long_audio_windows_2stage.py): Python calling hand-written HIP kernels through the C ABI of libzkast.so.
Importing this package never touches the GPU; the first call that needs it creates the context and fails loudly if
the library or a gfx950 device is missing.
"""
from .feature_extraction import ZkASTFeatureExtractor  # noqa: F401
from .modeling import ZkASTConfig, ZkASTForAudioClassification  # noqa: F401
from .pipeline import (SAMPLING_RATE, classify_recording, forward_probs,  # noqa: F401
                       classify_features, forward_probs_recording, load_audio, load_audio_to_device, load_stage_model, run_patient,
                       summarize_stage_outputs, window_audio, window_geometry)

__all__ = [
    "ZkASTFeatureExtractor", "ZkASTConfig", "ZkASTForAudioClassification", "SAMPLING_RATE",
    "classify_recording", "classify_features", "forward_probs", "forward_probs_recording", "load_audio", "load_audio_to_device",
    "load_stage_model",
    "run_patient", "summarize_stage_outputs", "window_audio", "window_geometry",
]
