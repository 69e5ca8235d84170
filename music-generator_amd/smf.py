"""Standard MIDI File reader/writer with the handful of python-midi names the DeepJ
codec touches (Pattern, Track, NoteOnEvent, NoteOffEvent, EndOfTrackEvent,
read_midifile, write_midifile).  python-midi (reference scripts/python.sh:11-14) is not
installable here, so this is an independent implementation of the SMF 1.0 wire format.
"""
import struct


class Pattern(list):
    def __init__(self, tracks=(), resolution=220, format=1):
        super().__init__(tracks)
        self.resolution = resolution
        self.format = format


class Track(list):
    pass


class Event:
    """Any event: only `tick` (delta time) matters to the decoder for non-note events."""
    status = None

    def __init__(self, tick=0, data=(), **kw):
        self.tick = tick
        self.data = list(data)

    def __repr__(self):
        return "%s(tick=%d, data=%r)" % (type(self).__name__, self.tick, self.data)


class _Note(Event):
    def __init__(self, tick=0, pitch=0, velocity=0, channel=0, data=None, **kw):
        if data is not None:
            pitch, velocity = data[0], data[1]
        super().__init__(tick, [pitch, velocity])
        self.channel = channel

    pitch = property(lambda s: s.data[0], lambda s, v: s.data.__setitem__(0, v))
    velocity = property(lambda s: s.data[1], lambda s, v: s.data.__setitem__(1, v))


class NoteOnEvent(_Note):
    status = 0x90


class NoteOffEvent(_Note):
    status = 0x80


class ChannelEvent(Event):
    def __init__(self, tick=0, status=0xB0, data=(), channel=0):
        super().__init__(tick, data)
        self.status = status
        self.channel = channel


class MetaEvent(Event):
    def __init__(self, tick=0, metacommand=0, data=()):
        super().__init__(tick, data)
        self.metacommand = metacommand


class EndOfTrackEvent(MetaEvent):
    def __init__(self, tick=0, **kw):
        super().__init__(tick, 0x2F, ())


class SysexEvent(Event):
    def __init__(self, tick=0, status=0xF0, data=()):
        super().__init__(tick, data)
        self.status = status


def _read_varlen(buf, pos):
    v = 0
    while True:
        b = buf[pos]
        pos += 1
        v = (v << 7) | (b & 0x7F)
        if not b & 0x80:
            return v, pos


def _write_varlen(v):
    out = [v & 0x7F]
    v >>= 7
    while v:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    return bytes(reversed(out))


_DATA_LEN = {0x80: 2, 0x90: 2, 0xA0: 2, 0xB0: 2, 0xC0: 1, 0xD0: 1, 0xE0: 2}


def _parse_track(buf):
    track = Track()
    pos, running = 0, None
    while pos < len(buf):
        tick, pos = _read_varlen(buf, pos)
        st = buf[pos]
        if st == 0xFF:
            kind = buf[pos + 1]
            n, pos = _read_varlen(buf, pos + 2)
            data = buf[pos:pos + n]
            pos += n
            if kind == 0x2F:
                track.append(EndOfTrackEvent(tick=tick))
                break
            track.append(MetaEvent(tick, kind, data))
            continue
        if st in (0xF0, 0xF7):
            n, pos = _read_varlen(buf, pos + 1)
            track.append(SysexEvent(tick, st, buf[pos:pos + n]))
            pos += n
            continue
        if st & 0x80:
            running = st
            pos += 1
        elif running is None:
            raise ValueError("data byte without running status")
        st = running
        n = _DATA_LEN[st & 0xF0]
        data = list(buf[pos:pos + n])
        pos += n
        hi, ch = st & 0xF0, st & 0x0F
        if hi == 0x90:
            track.append(NoteOnEvent(tick=tick, pitch=data[0], velocity=data[1], channel=ch))
        elif hi == 0x80:
            track.append(NoteOffEvent(tick=tick, pitch=data[0], velocity=data[1], channel=ch))
        else:
            track.append(ChannelEvent(tick, hi, data, ch))
    return track


def read_midifile(path_or_file):
    raw = path_or_file.read() if hasattr(path_or_file, "read") else open(path_or_file, "rb").read()
    if raw[:4] != b"MThd":
        raise ValueError("not a Standard MIDI File")
    hlen, = struct.unpack(">I", raw[4:8])
    fmt, ntrk, div = struct.unpack(">HHH", raw[8:14])
    if div & 0x8000:
        raise ValueError("SMPTE time division is not supported")
    pat = Pattern(resolution=div, format=fmt)
    pos = 8 + hlen
    while pos + 8 <= len(raw) and len(pat) < ntrk:
        tag = raw[pos:pos + 4]
        n, = struct.unpack(">I", raw[pos + 4:pos + 8])
        body = raw[pos + 8:pos + 8 + n]
        pos += 8 + n
        if tag == b"MTrk":
            pat.append(_parse_track(body))
    return pat


def _encode_event(e):
    head = _write_varlen(int(e.tick))
    if isinstance(e, MetaEvent):
        data = bytes(e.data)
        return head + bytes([0xFF, e.metacommand]) + _write_varlen(len(data)) + data
    if isinstance(e, SysexEvent):
        data = bytes(e.data)
        return head + bytes([e.status]) + _write_varlen(len(data)) + data
    ch = getattr(e, "channel", 0) & 0x0F
    return head + bytes([(e.status & 0xF0) | ch] + [int(d) & 0x7F for d in e.data])


def write_midifile(path_or_file, pattern):
    """Explicit status bytes (no running status), one MTrk per track -- the wire format of
    the reference's archived samples."""
    out = [b"MThd", struct.pack(">IHHH", 6, getattr(pattern, "format", 1), len(pattern), int(pattern.resolution))]
    for track in pattern:
        body = b"".join(_encode_event(e) for e in track)
        out += [b"MTrk", struct.pack(">I", len(body)), body]
    blob = b"".join(out)
    if hasattr(path_or_file, "write"):
        path_or_file.write(blob)
    else:
        with open(path_or_file, "wb") as f:
            f.write(blob)
