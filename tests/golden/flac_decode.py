"""FLAC decoding for the fixture generators: the product's own reader
(amt-saga_amd/amt_saga/flac.py).  Kept as a module so that
gen_golden_from_flac.py reads `import flac_decode as fd`."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(_ROOT, 'amt-saga_amd'))
from amt_saga.flac import decode, load_float  # noqa: E402,F401
