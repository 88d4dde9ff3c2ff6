"""Drop-in: ``from instrumentclassifier import InstrumentClassifier`` (training.py:25)."""
import _path  # noqa: F401
from amt_saga.heads import InstrumentClassifier  # noqa: F401,E402
