"""Drop-in for the reference's util_audio module (hot-path part):
``from util_audio import audio_complete`` as training.py:22 does.  The MIDI /
fluidsynth half of the reference module (note_sequence, render) is outside the
hot path and not provided; FLAC file I/O is (amt_saga.flac)."""
import _path  # noqa: F401
from amt_saga.audio import audio_complete  # noqa: F401,E402
from amt_saga import flac as _flac  # noqa: E402


def audio_from_file(audio_filename):
    """util_audio.py:962-964 (librosa.load(sr=None)): (float32 mono waveform, sample rate)."""
    y, sr = _flac.load_float(audio_filename)
    if y.ndim > 1:
        y = y.mean(axis=1)
    return y.astype('float32'), sr


def audio_to_flac(waveform, filename, sr=44100):
    """util_audio.py:966-968: save as PCM-24 FLAC."""
    _flac.save_float(waveform, filename, sr=sr, bps=24)
