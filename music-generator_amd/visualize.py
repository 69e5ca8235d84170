"""Style-embedding export for the TensorFlow embedding projector (http://projector.tensorflow.org/), the surface
of the reference's visualize.py:11-43: the `style` layer of the model applied to every one-hot style, written as
OUT_DIR/style_embedding_vec.tsv, and a two-column (Genre, Artist) label file OUT_DIR/style_embedding_labels.tsv with a
header row.  The layer runs on the GPU (dj_style_embedding); this file is host I/O only.

    python -m music_generator_amd.visualize
"""
import os

import numpy as np

from .constants import *  # noqa: F401,F403
from .util import build_or_load


def style_embeddings(models):
    """[NUM_STYLES, STYLE_UNITS]: get_layer('style') on the identity matrix (visualize.py:13-23)."""
    style_layer = models[0].get_layer('style')
    return np.asarray(style_layer(np.identity(NUM_STYLES)))


def style_labels():
    """Rows of the metadata file: header + one (genre, artist directory) pair per style, in style order
    (visualize.py:29-41; `styles` / `genre` are the nested lists of constants.py:10-40)."""
    rows = [['Genre', 'Artist']]
    for g, group in zip(genre, styles):
        rows.extend([g, artist] for artist in group)
    return np.array(rows)


def main(argv=None, models=None):
    models = models or build_or_load()
    print('Creating input')
    embedding = style_embeddings(models)
    print('Writing to out directory')
    os.makedirs(OUT_DIR, exist_ok=True)
    np.savetxt(os.path.join(OUT_DIR, 'style_embedding_vec.tsv'), embedding, delimiter='\t')
    np.savetxt(os.path.join(OUT_DIR, 'style_embedding_labels.tsv'), style_labels(), delimiter='\t', fmt='%s')
    return embedding


if __name__ == '__main__':
    main()
