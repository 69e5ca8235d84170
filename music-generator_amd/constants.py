"""Hyper-parameters of the DeepJ model -- same names and values as the reference's
constants.py (/root/reference/constants.py:1-84), so `from constants import *`
callers keep working after music_generator_amd.install()."""
import os

# Musical styles: genre -> composers; each composer directory under data/ is one style
# id, in this order (reference constants.py:4-42 spells the same 23 paths out).
_COMPOSERS = {
    'baroque': 'bach handel pachelbel',
    'classical': 'burgmueller clementi haydn beethoven brahms mozart',
    'romantic': ('balakirew borodin brahms chopin debussy liszt mendelssohn moszkowski mussorgsky '
                 'rachmaninov schubert schumann tchaikovsky tschai'),
}
genre = list(_COMPOSERS)
styles = [['data/%s/%s' % (g, c) for c in cs.split()] for g, cs in _COMPOSERS.items()]

NUM_STYLES = sum(len(s) for s in styles)          # 23

# MIDI
DEFAULT_RES = 96
MIDI_MAX_NOTES = 128
MAX_VELOCITY = 127

# Note range: 4 octaves from MIDI note 36
NUM_OCTAVES = 4
OCTAVE = 12
MIN_NOTE = 36
MAX_NOTE = MIN_NOTE + NUM_OCTAVES * OCTAVE
NUM_NOTES = MAX_NOTE - MIN_NOTE                    # 48

BEATS_PER_BAR = 4
NOTES_PER_BEAT = 4
NOTES_PER_BAR = NOTES_PER_BEAT * BEATS_PER_BAR     # 16

# Training
BATCH_SIZE = 16
SEQ_LEN = 8 * NOTES_PER_BAR                        # 128

# Model
OCTAVE_UNITS = 64
STYLE_UNITS = 64
NOTE_UNITS = 3
TIME_AXIS_UNITS = 256
NOTE_AXIS_UNITS = 128
TIME_AXIS_LAYERS = 2
NOTE_AXIS_LAYERS = 2

# Output locations.  The reference stores Keras HDF5 weights in out/model.h5; h5py is
# not available here, so weights are stored as .npz with Keras tensor names/layouts.
OUT_DIR = 'out'
MODEL_DIR = os.path.join(OUT_DIR, 'models')
MODEL_FILE = os.path.join(OUT_DIR, 'model.npz')
SAMPLES_DIR = os.path.join(OUT_DIR, 'samples')
CACHE_DIR = os.path.join(OUT_DIR, 'cache')
