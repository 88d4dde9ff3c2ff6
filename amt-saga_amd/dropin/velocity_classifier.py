"""Drop-in: ``from velocity_classifier import VelocityClassifier`` (training.py:28)."""
import _path  # noqa: F401
from amt_saga.heads import VelocityClassifier  # noqa: F401,E402
