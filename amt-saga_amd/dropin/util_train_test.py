"""Drop-in for util_train_test: Hyperparams, check_shape, list_to_nd_array and
the PATH_* constants main.py / training.py import."""
import _path  # noqa: F401
from amt_saga.hyperparams import (Hyperparams, check_shape, list_to_nd_array,  # noqa: F401,E402
                                  PATH_MODEL_META, PATH_NOTES, PATH_CHECKPOINTS)
