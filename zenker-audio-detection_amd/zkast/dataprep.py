"""Long-audio dataset preparation — the run-once step that DEFINES the on-disk input of the hot path
(utils/PrepareDatasetLongAudio.py:12-69): for every class except "Idle", every specimen folder `<id>_*`, the first
sub-folder whose name contains "long", every *.wav / *.WAV in it -> mono (channel mean, as librosa.load(mono=True)),
ORIGINAL sample rate (sr=None), written as `<long_root>/<class>/<specimen_id>/<name>.wav` (PCM_16, soundfile's default
for .wav).  librosa / soundfile are not needed: the RIFF reader / writer of pipeline.py are used.  Host-side only."""
from __future__ import annotations

import os

import numpy as np

from .pipeline import read_wav, write_wav_pcm16


def prepare_long_audio(raw_root: str, long_root: str, log=print) -> int:
    count = 0
    os.makedirs(long_root, exist_ok=True)
    for cl in os.listdir(raw_root):
        if cl == "Idle":
            continue
        os.makedirs(os.path.join(long_root, cl), exist_ok=True)
        for specimen in os.listdir(os.path.join(raw_root, cl)):
            specimen_id = specimen.split("_")[0]
            sdir = os.path.join(raw_root, cl, specimen)
            try:
                sub = [f for f in os.listdir(sdir) if os.path.isdir(os.path.join(sdir, f)) and "long" in f.lower()]
                files = os.listdir(os.path.join(sdir, sub[0]))
                wav_files = [f for f in files if (".wav" in f or ".WAV" in f)]
            except Exception as e:
                log(e)
                log("No long file for specimen: " + specimen + " in class: " + cl)
                continue
            for file in wav_files:
                filename, _ = os.path.splitext(file)
                wav, sr = read_wav(os.path.join(sdir, sub[0], file))
                mono = wav.mean(axis=0) if wav.shape[0] > 1 else wav[0]
                out_dir = os.path.join(long_root, cl, specimen_id)
                os.makedirs(out_dir, exist_ok=True)
                write_wav_pcm16(os.path.join(out_dir, filename + ".wav"), np.asarray(mono, np.float32), sr)
                count += 1
    log("Total files processed: " + str(count))
    return count
