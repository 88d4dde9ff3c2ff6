"""Drop-in: ``from timing_classifier import timming_classifier`` (training.py:27)."""
import _path  # noqa: F401
from amt_saga.heads import timming_classifier  # noqa: F401,E402
