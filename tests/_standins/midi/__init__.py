"""Build-owned stand-in for the absent third-party `python-midi` package.

Only the five classes the reference's codec touches (midi_util.py:14,17,41,50,92,
139,143,153) -- containers with .tick/.pitch/.velocity/.data -- so that the
reference's midi_encode/midi_decode can be imported to capture golden vectors
(SURVEY 8c).  Used only by tests/golden/make_golden.py and the codec tests.
"""


class Pattern(list):
    def __init__(self, tracks=(), resolution=220, format=1):
        super().__init__(tracks)
        self.resolution = resolution
        self.format = format


class Track(list):
    pass


class _Event:
    def __init__(self, tick=0, **kw):
        self.tick = tick


class _NoteEvent(_Event):
    def __init__(self, tick=0, pitch=0, velocity=0, **kw):
        super().__init__(tick)
        self.pitch = pitch
        self.velocity = velocity

    @property
    def data(self):
        return [self.pitch, self.velocity]


class NoteOnEvent(_NoteEvent):
    pass


class NoteOffEvent(_NoteEvent):
    pass


class EndOfTrackEvent(_Event):
    pass
