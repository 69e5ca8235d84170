"""Stand-in for the absent third-party `python-midi` package, used ONLY so that the
reference's codec can be imported to capture golden vectors (SURVEY 8c,
tests/golden/make_golden.py).  It re-exports this build's own SMF module, which has the
five class names and the two I/O functions the reference touches (midi_util.py:14,17,41,
50,92,139,143,153,194; generate.py:134)."""
import os
import sys

_root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if _root not in sys.path:
    sys.path.append(_root)
from music_generator_amd.smf import *  # noqa: E402,F401,F403
from music_generator_amd.smf import (EndOfTrackEvent, NoteOffEvent, NoteOnEvent, Pattern, Track,  # noqa: E402,F401
                                     read_midifile, write_midifile)
