"""Style-embedding export for the TensorFlow embedding projector (reference visualize.py:11-43):
writes the learned style vectors (one row per composer style) and a label file as TSV.

    python -m music_generator_amd.visualize [--out out/projector]
"""
import argparse
import os

import numpy as np

from .constants import *  # noqa: F401,F403
from .util import build_or_load, one_hot


def style_embeddings(models):
    """[NUM_STYLES, STYLE_UNITS]: the `style` Dense layer applied to the identity (visualize.py:13-17)."""
    layer = models[0].get_layer('style')
    return layer(np.array([one_hot(i, NUM_STYLES) for i in range(NUM_STYLES)]))


def main(argv=None):
    ap = argparse.ArgumentParser(description='Exports the style embedding matrix as TSV.')
    ap.add_argument('--out', default=os.path.join(OUT_DIR, 'projector'))
    args = ap.parse_args(argv)
    models = build_or_load()
    emb = style_embeddings(models)
    os.makedirs(args.out, exist_ok=True)
    np.savetxt(os.path.join(args.out, 'style_embeddings.tsv'), emb, delimiter='\t')
    with open(os.path.join(args.out, 'style_labels.tsv'), 'w') as f:
        for group in styles:
            for path in group:
                f.write(os.path.basename(path) + '\n')
    print('Wrote', args.out)


if __name__ == '__main__':
    main()
