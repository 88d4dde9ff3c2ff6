"""Puts amt-saga_amd/ on sys.path so the drop-in modules find the amt_saga package."""
import os
import sys

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)
