"""Small helpers with the reference's names (reference util.py:8-33)."""
import os

import numpy as np

from .constants import *  # noqa: F401,F403


def one_hot(i, nb_classes):
    """Length-`nb_classes` float64 vector with a 1 at position i (util.py:8-11)."""
    v = np.zeros((nb_classes,))
    v[i] = 1
    return v


def build_or_load(allow_load=True, **build_kwargs):
    """Build the three models with default settings, print the summary and try to
    restore MODEL_FILE; any failure to load is reported and ignored, exactly like the
    reference's bare `except` (util.py:13-23)."""
    from .model import KerasHDF5Error, build_models
    models = build_models(**build_kwargs)
    models[0].summary()
    if allow_load:
        # the reference's own checkpoint name (out/model.h5) is looked at too: a Keras HDF5 file there must not be
        # ignored silently -- load_weights names the converter
        legacy = os.path.splitext(MODEL_FILE)[0] + '.h5'
        try:
            models[0].load_weights(MODEL_FILE if os.path.exists(MODEL_FILE) or not os.path.exists(legacy) else legacy)
            print('Loaded model from file.')
        except KerasHDF5Error:
            raise
        except Exception:
            print('Unable to load model from file.')
    return models


def get_all_files(paths):
    """Every `*.mid` file below the given directories, os.walk order (util.py:25-33)."""
    found = []
    for top in paths:
        for root, _dirs, files in os.walk(top):
            found.extend(os.path.join(root, f) for f in files
                         if f.endswith('.mid') and os.path.isfile(os.path.join(root, f)))
    return found
