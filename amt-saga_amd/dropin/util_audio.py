"""Drop-in for the reference's util_audio module (hot-path part):
``from util_audio import audio_complete`` as training.py:22 does.  The MIDI /
fluidsynth / file-I/O half of the reference module (note_sequence, audio_to_flac,
...) is outside the hot path and not provided."""
import _path  # noqa: F401
from amt_saga.audio import audio_complete  # noqa: F401,E402
