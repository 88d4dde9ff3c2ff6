"""Drop-in: ``from pitch_classifier import pitch_classifier`` (training.py:26)."""
import _path  # noqa: F401
from amt_saga.heads import pitch_classifier  # noqa: F401,E402
