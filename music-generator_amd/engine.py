"""Host-side engine: owns device memory (through torch) and calls the C ABI.

PyTorch is plumbing here -- tensor allocation, streams, torch.distributed -- every
arithmetic op of the hot path runs inside libdeepj_hip.so.
"""
from __future__ import annotations

import ctypes as C
import math
import weakref
from dataclasses import dataclass

import numpy as np
import torch

from . import _lib
from . import constants as K


@dataclass(frozen=True)
class DeepJConfig:
    """Model hyper-parameters; defaults = reference constants.py:42-77."""
    num_notes: int = K.NUM_NOTES
    time_steps: int = K.SEQ_LEN
    num_styles: int = K.NUM_STYLES
    notes_per_bar: int = K.NOTES_PER_BAR
    octave: int = K.OCTAVE
    octave_units: int = K.OCTAVE_UNITS
    style_units: int = K.STYLE_UNITS
    note_units: int = K.NOTE_UNITS
    time_axis_units: int = K.TIME_AXIS_UNITS
    note_axis_units: int = K.NOTE_AXIS_UNITS
    time_axis_layers: int = K.TIME_AXIS_LAYERS
    note_axis_layers: int = K.NOTE_AXIS_LAYERS
    recurrent_activation: str = "hard_sigmoid"     # Keras 2.x LSTM default
    dtype: str = "f32"                             # "f32" (parity) | "bf16" (throughput)

    def cstruct(self, batch, time_steps=None, input_dropout=0.0, dropout=0.0, kernel_flags=0, fuse_xw_min_tiles=0):
        c = _lib.DjConfig()
        c.kernel_flags = int(kernel_flags)               # include/deepj_hip.h DJ_KF_* (per engine)
        c.fuse_xw_min_tiles = int(fuse_xw_min_tiles)
        c.batch = int(batch)
        c.time_steps = int(self.time_steps if time_steps is None else time_steps)
        c.num_notes = self.num_notes
        c.num_styles = self.num_styles
        c.notes_per_bar = self.notes_per_bar
        c.octave = self.octave
        c.octave_units = self.octave_units
        c.style_units = self.style_units
        c.note_units = self.note_units
        c.time_axis_units = self.time_axis_units
        c.note_axis_units = self.note_axis_units
        c.time_axis_layers = self.time_axis_layers
        c.note_axis_layers = self.note_axis_layers
        c.dtype = {"f32": _lib.DTYPE_F32, "bf16": _lib.DTYPE_BF16}[self.dtype]
        c.recurrent_sigmoid = {"hard_sigmoid": 0, "sigmoid": 1}[self.recurrent_activation]
        c.input_dropout = float(input_dropout)
        c.dropout = float(dropout)
        return c


def param_layout(cfg: DeepJConfig):
    """[(name, offset, shape)] straight from the library (dj_param_info)."""
    lib = _lib.load()
    c = cfg.cstruct(1)
    out = []
    name = C.create_string_buffer(64)
    off = C.c_int64()
    shape = (C.c_int32 * 4)()
    nd = C.c_int32()
    i = 0
    while True:
        rc = lib.dj_param_info(C.byref(c), i, name, 64, C.byref(off), shape, C.byref(nd))
        if rc == 1000:
            break
        _lib.check(rc, "dj_param_info")
        out.append((name.value.decode(), int(off.value), tuple(int(shape[k]) for k in range(nd.value))))
        i += 1
    return out


def param_count(cfg: DeepJConfig) -> int:
    n = _lib.load().dj_param_count(C.byref(cfg.cstruct(1)))
    if n < 0:
        raise _lib.DeepJError("dj_param_count: invalid configuration")
    return int(n)


def init_params_numpy(cfg: DeepJConfig, seed: int = 1234) -> np.ndarray:
    """Keras default initialisers: glorot-uniform kernels, orthogonal
    recurrent kernels, zero biases with unit forget-gate bias.  Flat fp32 vector."""
    rs = np.random.RandomState(seed)
    flat = np.zeros(param_count(cfg), np.float32)
    for name, off, shape in param_layout(cfg):
        n = int(np.prod(shape))
        if name.endswith("/bias"):
            w = np.zeros(shape, np.float32)
            if "lstm" in name:
                h = shape[0] // 4
                w[h:2 * h] = 1.0
        elif name.endswith("recurrent_kernel"):
            a = rs.normal(0.0, 1.0, shape)
            u, _, v = np.linalg.svd(a, full_matrices=False)
            w = (u if u.shape == shape else v).reshape(shape).astype(np.float32)
        else:
            if len(shape) == 3:
                fan_in, fan_out = shape[0] * shape[1], shape[0] * shape[2]
            else:
                fan_in, fan_out = shape[0], shape[1]
            lim = math.sqrt(6.0 / (fan_in + fan_out))
            w = rs.uniform(-lim, lim, shape).astype(np.float32)
        flat[off:off + n] = w.ravel()
    return flat


def _dev_f32(a, device):
    if isinstance(a, torch.Tensor):
        return a.to(device=device, dtype=torch.float32).contiguous()
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).to(device)


def _stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


# Every observation of cluster faults by the host, on any path (census after predict / generation, the [loss, faults]
# read-back of a training step): process-wide, empty in a healthy run.  The GPU tests fail on any entry that a test did
# not inject (tests/conftest.py), so a recurrence on the driver's box cannot hide behind a fallback.
FAULT_LOG: list = []
_LIVE = weakref.WeakSet()          # engines alive in this process
# Stall census of the engines that have been released (Engine.close(), or drain_pending() for those that died unclosed,
# read the fault line once more): how many there
# were, the most waits any of them saw with polls > 2^20 cycles apart, the longest gap between two polls of a wait.
CENSUS = {"engines": 0, "stalled_waits": 0, "max_poll_gap_cycles": 0, "backward_clock_polls": 0}

# Workspaces of engines that were garbage-collected without close(): (lib, cfg, tensor, pointer, bytes, device, meta).
# Engine.__del__ only appends here (no HIP work in a finalizer); drain_pending() takes their census at a safe point.
_PENDING: list = []


def _census_of_workspace(lib, c, ws_ptr, ws_bytes, device, log):
    """Stall census -> CENSUS, unread fault counts -> FAULT_LOG (through `log(what, report, events)`), for a workspace
    that is about to be released.  Blocking (one copy of the fault line on the current stream)."""
    words = (C.c_int32 * _lib.FAULT_REPORT_WORDS)()
    try:
        with torch.cuda.device(device):
            rc = lib.dj_workspace_cluster_fault_report(C.byref(c), ws_ptr, ws_bytes, words, _stream_ptr())
    except Exception:
        return
    if rc != 0:
        return
    rep = decode_fault_report(words)
    CENSUS["engines"] += 1
    CENSUS["stalled_waits"] = max(CENSUS["stalled_waits"], rep["stalled_waits"])
    CENSUS["max_poll_gap_cycles"] = max(CENSUS["max_poll_gap_cycles"], rep["max_poll_gap_cycles"])
    CENSUS["backward_clock_polls"] = max(CENSUS["backward_clock_polls"], rep["backward_clock_polls"])
    if rep["expired"] or rep["misplaced"]:
        log("unread when the engine was destroyed", rep, rep["expired"] + rep["misplaced"])


def drain_pending():
    """Census of the engines that died without close() (see Engine.__del__), then their workspaces are freed.  Called
    from Engine(), Engine.close(), the test fixture and bench.py; does nothing while the current stream is being
    captured into a graph (the copy would invalidate the capture) -- the entries wait for the next call."""
    if not _PENDING:
        return 0
    try:
        if torch.cuda.is_current_stream_capturing():
            return 0
    except Exception:
        return 0
    n = 0
    while _PENDING:
        lib, c, ws, ws_ptr, ws_bytes, device, meta = _PENDING.pop()

        def log(what, rep, events, meta=meta):
            FAULT_LOG.append(dict(rep, what=what, events=int(events), **meta))
        _census_of_workspace(lib, c, ws_ptr, ws_bytes, device, log)
        del ws
        n += 1
    return n


_WAIT_KINDS = {1: "bf16 sweep", 2: "bf16 cooperative body", 3: "fp32 inference sweep",
               4: "bf16 sweep, tagged h fragments (counter = fragments that had arrived)", 5: "pair BPTT (tools)",
               7: "bf16 cooperative body, tagged h fragments (counter = fragments that had arrived)",
               6: "two-tile BPTT (tools)"}


def decode_fault_report(words):
    """dj_workspace_cluster_fault_report words -> dict (include/deepj_hip.h DJ_FAULT_REPORT_WORDS)."""
    w = [int(v) for v in words]
    rep = {"expired": w[0], "misplaced": w[1], "hook": w[2], "stalled_waits": w[4], "max_poll_gap_cycles": w[5] << 10, "backward_clock_polls": w[6],
           "first_expired": None}
    if w[8]:
        who = w[9] & 0xFFFFFFFF
        rep["first_expired"] = {
            "kernel": _WAIT_KINDS.get((who >> 24) & 15, "kind %d" % ((who >> 24) & 15)),
            "producer_counter": bool(who & (1 << 28)), "cluster": (who >> 12) & 0xFFF, "member": (who >> 8) & 15,
            "wave": who & 255, "step": w[10], "counter_seen": w[11], "target": w[12], "polls": w[13],
            "elapsed_cycles": (w[14] & 0xFFFFFFFF) | ((w[15] & 0xFFFFFFFF) << 32),
            "max_poll_gap_cycles": w[16] & 0xFFFFFFFF, "xcc": w[17] - 1}
    return rep


def describe_fault_report(rep):
    txt = "%d expired waits (a member of a cluster never arrived: device shared with other kernels), %d workgroups of " \
          "clusters that were not dealt round-robin over the XCDs" % (rep["expired"], rep["misplaced"])
    f = rep.get("first_expired")
    if f:
        txt += ("; first expired wait: %s, cluster %d member %d wave %d (XCC %d) at step %d%s: counter %d of %d after "
                "%d polls / %d shader cycles, longest gap between two polls %d cycles"
                % (f["kernel"], f["cluster"], f["member"], f["wave"], f["xcc"], f["step"],
                   " on the producing layer's counter" if f["producer_counter"] else "", f["counter_seen"], f["target"],
                   f["polls"], f["elapsed_cycles"], f["max_poll_gap_cycles"]))
    if rep.get("stalled_waits") or rep.get("backward_clock_polls"):
        txt += ("; stall census of this workspace: %d waits with polls > 2^20 cycles apart, longest gap %d cycles, %d polls "
                "behind a shader clock that had gone backwards (wave restored on another XCC)"
                % (rep["stalled_waits"], rep["max_poll_gap_cycles"], rep.get("backward_clock_polls", 0)))
    return txt


class Engine:
    """One (config, batch, time_steps) instance = one workspace in HBM."""

    def __init__(self, cfg: DeepJConfig, batch: int, time_steps: int | None = None, device="cuda:0",
                 input_dropout: float = 0.0, dropout: float = 0.0, kernel_flags: int = 0, fuse_xw_min_tiles: int = 0):
        self._closed = True                              # until the workspace exists (see __del__)
        self.ws = None
        if not torch.cuda.is_available():
            raise _lib.DeepJError("no HIP device visible: the DeepJ engine has no CPU path")
        drain_pending()                                  # workspaces of engines that died unclosed: census, then free
        self.lib = _lib.load()
        self.cfg = cfg
        self.device = torch.device(device)
        self.batch = int(batch)
        self.time_steps = int(cfg.time_steps if time_steps is None else time_steps)
        self.c = cfg.cstruct(batch, self.time_steps, input_dropout, dropout, kernel_flags, fuse_xw_min_tiles)
        self.flags_epoch = 0                             # bumped when kernel_flags change (captured graphs are stale then)
        self.nparams = int(self.lib.dj_param_count(C.byref(self.c)))
        nbytes = int(self.lib.dj_workspace_bytes(C.byref(self.c)))
        if self.nparams < 0 or nbytes < 0:
            raise _lib.DeepJError("invalid DeepJ configuration for the HIP engine")
        self.ws_bytes = nbytes
        with torch.cuda.device(self.device):
            self.ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=self.device)
            pad = (-self.ws.data_ptr()) % 256
            self.ws_ptr = C.c_void_p(self.ws.data_ptr() + pad)
            _lib.check(self.lib.dj_workspace_init(C.byref(self.c), self.ws_ptr, nbytes, _stream_ptr()),
                       "dj_workspace_init")
        # [mean loss, cluster faults of the call, of which expired waits, misplaced]: one device-to-host copy serves
        # all of it (dj_workspace_faults_async)
        self.loss = torch.zeros(4, dtype=torch.float32, device=self.device)
        self.last_fault = None                           # the newest FAULT_LOG entry of this engine
        self.fault_of_last_take = None                   # ... and the one the last take_async_faults() made, if any
        self._closed = False
        _LIVE.add(self)

    def close(self):
        """Last look at the fault line, then release the workspace: the stall census goes to the module's CENSUS, counts
        nobody read to FAULT_LOG.  Idempotent.  This is the only place where a dying engine does HIP work -- call it
        from the owner (Model.close, bench.py, the test fixture), never from a finalizer."""
        if self._closed:
            return
        self._closed = True
        _LIVE.discard(self)
        _census_of_workspace(self.lib, self.c, self.ws_ptr, self.ws_bytes, self.device, self._log_faults)
        self.ws = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        # NO HIP call here: a finalizer runs wherever the collector does -- on another thread, inside somebody's
        # torch.cuda.graph capture (a synchronise there invalidates the capture).  An engine that dies unclosed only
        # hands its workspace to _PENDING; drain_pending() reads the fault line at the next safe point (Engine(),
        # close(), the test fixture, bench.py).
        try:
            if not getattr(self, "_closed", True) and self.ws is not None:
                self._closed = True
                _PENDING.append((self.lib, self.c, self.ws, self.ws_ptr, self.ws_bytes, self.device, self._fault_meta()))
        except Exception:                                # interpreter shutting down
            pass

    def set_kernel_flags(self, flags: int):
        """Per-engine kernel selection (DJ_KF_*); e.g. KF_NO_CLUSTER after a cluster fault.  Affects later calls of
        THIS engine only -- no environment variable, no other engine."""
        if int(flags) != int(self.c.kernel_flags):
            self.c.kernel_flags = int(flags)
            self.flags_epoch += 1

    def cluster_fault_report(self):
        """The fault line of this engine's workspace, decoded (decode_fault_report): counts, stall census, description of
        the first expired wait.  Resets nothing; drains the current stream."""
        words = (C.c_int32 * _lib.FAULT_REPORT_WORDS)()
        with torch.cuda.device(self.device):
            _lib.check(self.lib.dj_workspace_cluster_fault_report(C.byref(self.c), self.ws_ptr, self.ws_bytes, words,
                                                                  _stream_ptr()), "dj_workspace_cluster_fault_report")
        return decode_fault_report(words)

    def _fault_meta(self):
        return dict(kernel_flags=int(self.c.kernel_flags), batch=self.batch, time_steps=self.time_steps, dtype=self.cfg.dtype)

    def _log_faults(self, what, rep, n):
        entry = dict(rep, what=what, events=int(n), **self._fault_meta())
        self.last_fault = entry
        FAULT_LOG.append(entry)
        return entry

    def cluster_faults(self, what="census") -> int:
        """Events recorded by the weight-stationary cluster kernels in THIS engine's workspace since the last
        call (expired waits, clusters spread over several XCDs; include/deepj_hip.h dj_workspace_cluster_faults_take).
        Zero in a healthy run; non-zero means the affected tiles -- and the loss -- are NaN, and the observation is
        appended to FAULT_LOG with the description of the first expired wait.  ONE blocking round trip: the fault line
        comes back with the count, the reset is queued behind it on the stream.  Drains the current stream."""
        words = (C.c_int32 * _lib.FAULT_REPORT_WORDS)()
        with torch.cuda.device(self.device):
            n = int(self.lib.dj_workspace_cluster_faults_take(C.byref(self.c), self.ws_ptr, self.ws_bytes, words,
                                                              _stream_ptr()))
        if n < 0:
            raise _lib.DeepJError("dj_workspace_cluster_faults_take failed")
        if n:
            self._log_faults(what, decode_fault_report(words), n)
        return n + self.take_async_faults(what=what)

    def take_async_faults(self, value=None, what="training step") -> int:
        """Faults that train_fwd_bwd calls have added to loss[1:4] (dj_workspace_faults_async) since the last take;
        `value`: the number if the caller has already read loss[1] with the loss.  `self.fault_of_last_take` is the
        FAULT_LOG entry this call made, or None when THIS call saw nothing (so that nobody describes a stale
        `last_fault` as this step's)."""
        m = int(self.loss[1].item()) if value is None else int(value)
        self.fault_of_last_take = None
        if m:
            host = self.loss.cpu().numpy()
            words = (C.c_int32 * _lib.FAULT_REPORT_WORDS)()
            with torch.cuda.device(self.device):         # the async census left the first wait's description in place:
                self.lib.dj_workspace_cluster_faults_take(C.byref(self.c), self.ws_ptr, self.ws_bytes, words,
                                                          _stream_ptr())     # ... the host takes (and resets) it here
            rep = decode_fault_report(words)
            rep["expired"], rep["misplaced"] = int(host[2]), int(host[3])
            self.loss[1:].zero_()
            self.fault_of_last_take = self._log_faults(what, rep, m)
        return m

    def raise_on_cluster_faults(self, what):
        n = self.cluster_faults(what)
        if n:
            raise _lib.DeepJError(
                "%s: %d cluster faults in the recurrent forward kernel (%s); the results are NaN.  "
                "Set DEEPJ_CLUSTER=0 to use the per-tile kernel." % (what, n, describe_fault_report(self.last_fault)))

    # -- shapes
    def _shapes(self):
        B, T, N = self.batch, self.time_steps, self.cfg.num_notes
        return (B, T, N, 3), (B, T, self.cfg.notes_per_bar), (B, T, self.cfg.num_styles)

    def _check(self, t, shape, what):
        if tuple(t.shape) != tuple(shape):
            raise ValueError(f"{what}: expected shape {tuple(shape)}, got {tuple(t.shape)}")

    def pitch_bins(self, notes_full, seed=0, train=True, batch_offset=0):
        """pitch_bins table [octave, B_full, T] of a WHOLE batch (model.py:43-45; dj_pitch_bins): what the micro-batches
        of that batch read through train_fwd_bwd(..., full_batch=, batch_offset=, bins_full=).  batch_offset: these
        samples are rows batch_offset... of a larger (global) batch -- a data-parallel rank's part of the table."""
        bfull, T = int(notes_full.shape[0]), self.time_steps
        self._check(notes_full, (bfull, T, self.cfg.num_notes, self.cfg.note_units), "notes_full")
        cfull = self.cfg.cstruct(bfull, T, self.c.input_dropout, self.c.dropout, self.c.kernel_flags)
        bins = torch.empty((self.cfg.octave, bfull, T), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.dj_pitch_bins(C.byref(cfull), _lib.ptr(notes_full), _lib.ptr(bins),
                                              C.c_uint64(int(seed) & (2 ** 64 - 1)), 1 if train else 0,
                                              int(batch_offset), _stream_ptr()),
                       "dj_pitch_bins")
        return bins

    def train_fwd_bwd(self, params, grads, notes, chosen, beat, style, target, seed=0, out=None, accumulate=False,
                      full_batch=0, batch_offset=0, bins_full=None):
        """Forward + BPTT.  All arguments are device fp32 tensors; returns the loss
        tensor (device, shape [1]); grads is overwritten, or added to when `accumulate`
        (gradient accumulation over micro-batches).  With `bins_full` (pitch_bins of the whole batch) the call is
        samples [batch_offset, batch_offset + B) of a batch of `full_batch`, exactly (dj_train_fwd_bwd_mb)."""
        s3, sb, ss = self._shapes()
        for t, sh, nm in ((notes, s3, "notes"), (chosen, s3, "chosen"), (target, s3, "target"),
                          (beat, sb, "beat"), (style, ss, "style")):
            self._check(t, sh, nm)
        assert params.numel() == self.nparams and grads.numel() == self.nparams
        if bins_full is not None:
            self._check(bins_full, (self.cfg.octave, int(full_batch), self.time_steps), "bins_full")
        with torch.cuda.device(self.device):
            rc = self.lib.dj_train_fwd_bwd_mb(C.byref(self.c), _lib.ptr(params), _lib.ptr(grads), _lib.ptr(notes),
                                              _lib.ptr(chosen), _lib.ptr(beat), _lib.ptr(style), _lib.ptr(target),
                                              _lib.ptr(out), _lib.ptr(self.loss), self.ws_ptr, self.ws_bytes,
                                              C.c_uint64(int(seed) & (2 ** 64 - 1)), 1 if accumulate else 0,
                                              int(full_batch) if bins_full is not None else 0,
                                              int(batch_offset) if bins_full is not None else 0, _lib.ptr(bins_full),
                                              _stream_ptr())
            _lib.check(rc, "dj_train_fwd_bwd")
            # fault census of this call next to the loss, without a host round trip
            _lib.check(self.lib.dj_workspace_faults_async(C.byref(self.c), self.ws_ptr, self.ws_bytes,
                                                          C.c_void_p(self.loss.data_ptr() + 4), _stream_ptr()),
                       "dj_workspace_faults_async")
        return self.loss

    def style_embedding(self, params, style_in):
        """style Dense layer (model.py:141-142) on style_in [rows, num_styles] -> [rows, style_units] (device)."""
        rows = int(style_in.shape[0])
        self._check(style_in, (rows, self.cfg.num_styles), "style_in")
        out = torch.empty((rows, self.cfg.style_units), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self.lib.dj_style_embedding(C.byref(self.c), _lib.ptr(params), _lib.ptr(style_in), rows,
                                                   _lib.ptr(out), _stream_ptr()), "dj_style_embedding")
        return out

    def predict(self, params, notes, chosen, beat, style, target=None):
        s3, sb, ss = self._shapes()
        for t, sh, nm in ((notes, s3, "notes"), (chosen, s3, "chosen"), (beat, sb, "beat"), (style, ss, "style")):
            self._check(t, sh, nm)
        out = torch.empty(s3, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            rc = self.lib.dj_predict(C.byref(self.c), _lib.ptr(params), _lib.ptr(notes), _lib.ptr(chosen),
                                     _lib.ptr(beat), _lib.ptr(style), _lib.ptr(target),
                                     _lib.ptr(out), _lib.ptr(self.loss) if target is not None else None,
                                     self.ws_ptr, self.ws_bytes, _stream_ptr())
        _lib.check(rc, "dj_predict")
        return (out, self.loss[:1]) if target is not None else out

    def time_model_predict(self, params, notes, beat, style):
        s3, sb, ss = self._shapes()
        self._check(notes, s3, "notes"); self._check(beat, sb, "beat"); self._check(style, ss, "style")
        B, T, N = self.batch, self.time_steps, self.cfg.num_notes
        out = torch.empty((B, T, N, self.cfg.time_axis_units), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            rc = self.lib.dj_time_model_predict(C.byref(self.c), _lib.ptr(params), _lib.ptr(notes), _lib.ptr(beat),
                                                _lib.ptr(style), _lib.ptr(out), self.ws_ptr, self.ws_bytes,
                                                _stream_ptr())
        _lib.check(rc, "dj_time_model_predict")
        return out

    def note_model_predict(self, params, features, chosen, style):
        s3, _, ss = self._shapes()
        B, T, N = self.batch, self.time_steps, self.cfg.num_notes
        self._check(features, (B, T, N, self.cfg.time_axis_units), "features")
        self._check(chosen, s3, "chosen"); self._check(style, ss, "style")
        out = torch.empty(s3, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            rc = self.lib.dj_note_model_predict(C.byref(self.c), _lib.ptr(params), _lib.ptr(features),
                                                _lib.ptr(chosen), _lib.ptr(style), _lib.ptr(out), self.ws_ptr,
                                                self.ws_bytes, _stream_ptr())
        _lib.check(rc, "dj_note_model_predict")
        return out


    def generate_step(self, params, notes_win, beat_win, style_win, uniforms, temperature):
        """One generated time step for all `batch` pieces (dj_generate_step).  uniforms:
        float64 device tensor [2*N*G]; temperature: float32 device tensor [G].
        Returns (next_notes [G,N,3] float32 device, draws_used int32 device [2] = draws consumed, near ties)."""
        s3, sb, ss = self._shapes()
        self._check(notes_win, s3, "notes"); self._check(beat_win, sb, "beat"); self._check(style_win, ss, "style")
        G, N = self.batch, self.cfg.num_notes
        assert uniforms.dtype == torch.float64 and uniforms.numel() >= 2 * N * G
        assert temperature.dtype == torch.float32 and temperature.numel() == G
        out = torch.empty((G, N, 3), dtype=torch.float32, device=self.device)
        used = torch.zeros(2, dtype=torch.int32, device=self.device)
        with torch.cuda.device(self.device):
            rc = self.lib.dj_generate_step(C.byref(self.c), _lib.ptr(params), _lib.ptr(notes_win), _lib.ptr(beat_win),
                                           _lib.ptr(style_win), _lib.ptr(uniforms), _lib.ptr(temperature),
                                           _lib.ptr(out), _lib.ptr(used), self.ws_ptr, self.ws_bytes, _stream_ptr())
        _lib.check(rc, "dj_generate_step")
        return out, used


GEN_STATE_DTYPE = np.dtype([("step", "<i4"), ("draw_off", "<i4"), ("near_ties", "<i4"), ("first_near_step", "<i4"),
                            ("temperature", "<f8", (8,)), ("default_temp", "<f8", (8,)), ("silent", "<i4", (8,))])


class ResidentGeneration:
    """Device-resident sampling run (dj_generate_prepare + dj_generate_step_prepared): windows, temperature schedule,
    RNG-draw offset and emitted notes live in HBM; two ping-pong steps are captured into one
    hipGraph (through torch.cuda.CUDAGraph) and replayed.  `run(k, uniforms)` advances k steps."""

    def __init__(self, engine: Engine, params, styles, default_temp=1.0, steps_cap=4096, use_graph=True, prepared=True):
        self.e, self.params = engine, params
        self.prepared = prepared          # False: dj_generate_step_resident, which redoes the per-run constants every step
        G, T, N = engine.batch, engine.time_steps, engine.cfg.num_notes
        dev = engine.device
        assert engine.lib.dj_gen_state_size() == GEN_STATE_DTYPE.itemsize
        z = lambda *sh: torch.zeros(sh, dtype=torch.float32, device=dev)
        self.notes = [z(G, T, N, 3), z(G, T, N, 3)]
        self.beat = [z(G, T, engine.cfg.notes_per_bar), z(G, T, engine.cfg.notes_per_bar)]
        st = np.asarray(styles, np.float32)                       # [G, S], repeated over the window
        self.style = torch.from_numpy(np.repeat(st[:, None, :], T, axis=1).copy()).to(dev)
        self.results = z(steps_cap, G, N, 3)
        self.cap = steps_cap
        host = np.zeros(1, GEN_STATE_DTYPE)
        host["temperature"][0, :G] = default_temp
        host["default_temp"][0, :G] = default_temp
        host["silent"][0, :G] = engine.cfg.notes_per_bar          # generate.py:24
        host["first_near_step"] = -1
        self.state = torch.from_numpy(host.view(np.uint8).copy()).to(dev)
        self.pool = torch.zeros(2 * N * G * 128, dtype=torch.float64, device=dev)
        self.cur = 0                                              # which window buffer is current
        self.graph = None
        self._graph_epoch = engine.flags_epoch
        self.last_state = None
        self._want_graph = use_graph

    def _prepare(self):
        """Per-run constants (packed weights, style terms, ...) into the workspace: dj_generate_prepare.  Every run()
        starts with it, so other uses of the engine between two run() calls are fine."""
        e = self.e
        if not self.prepared:
            return
        with torch.cuda.device(e.device):
            rc = e.lib.dj_generate_prepare(C.byref(e.c), _lib.ptr(self.params), _lib.ptr(self.style), e.ws_ptr, e.ws_bytes,
                                           _stream_ptr())
        _lib.check(rc, "dj_generate_prepare")

    def _step(self, src):
        e = self.e
        fn = e.lib.dj_generate_step_prepared if self.prepared else e.lib.dj_generate_step_resident
        with torch.cuda.device(e.device):
            rc = fn(
                C.byref(e.c), _lib.ptr(self.params), _lib.ptr(self.state), _lib.ptr(self.results), _lib.ptr(self.pool),
                _lib.ptr(self.notes[src]), _lib.ptr(self.notes[1 - src]), _lib.ptr(self.beat[src]),
                _lib.ptr(self.beat[1 - src]), _lib.ptr(self.style), e.ws_ptr, e.ws_bytes, _stream_ptr())
        _lib.check(rc, "dj_generate_step_prepared")

    def read_state(self):
        return self.state.cpu().numpy().view(GEN_STATE_DTYPE)[0]

    def _set_draw_off(self, v):
        host = self.read_state().copy()
        host["draw_off"] = v
        self.state.copy_(torch.from_numpy(np.array([host]).view(np.uint8).reshape(-1)))

    def run(self, k, uniforms):
        """Advance k time steps consuming `uniforms` (float64, >= 2*N*G*k values, reference
        draw order).  Returns (notes [k,G,N,3] float32 numpy, draws consumed)."""
        G, N = self.e.batch, self.e.cfg.num_notes
        assert len(uniforms) >= 2 * N * G * k and len(uniforms) <= self.pool.numel()
        st0 = int(self.read_state()["step"])
        assert st0 + k <= self.cap
        self.pool[:len(uniforms)].copy_(torch.as_tensor(np.asarray(uniforms, np.float64)))
        self._set_draw_off(0)
        if self._graph_epoch != self.e.flags_epoch:               # kernel selection changed: the capture is stale
            self.graph, self._graph_epoch = None, self.e.flags_epoch
        self._prepare()
        done = 0
        if self._want_graph and self.graph is None and self.cur == 0 and k >= 2:
            try:                                                  # capture two ping-pong steps once
                self._step(0); self._step(1)                      # warm-up (also one-time function attributes)
                done = 2
                torch.cuda.synchronize(self.e.device)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    self._step(0); self._step(1)
                self.graph = g
            except Exception:
                self.graph = None
                self._want_graph = False
                torch.cuda.synchronize(self.e.device)
        while done < k:
            if self.graph is not None and self.cur == 0 and k - done >= 2:
                self.graph.replay()
                done += 2
            else:
                self._step(self.cur)
                self.cur ^= 1
                done += 1
        out = self.results[st0:st0 + k].cpu().numpy()
        # a cluster fault (expired wait, members of a cluster on different XCDs) poisons the affected rows with NaN,
        # which the sampler would turn into silence: never hand such notes out
        self.e.raise_on_cluster_faults("generation")
        self.last_state = self.read_state()                       # schedule after this chunk (checked by the host mirror)
        return out, int(self.last_state["draw_off"])


class Nadam:
    """Keras-2 Nadam state (model.py:152) around dj_nadam_step."""

    def __init__(self, nparams, device, lr=0.002, beta_1=0.9, beta_2=0.999, epsilon=1e-8, schedule_decay=0.004):
        self.lib = _lib.load()
        self.lr, self.beta_1, self.beta_2, self.epsilon, self.schedule_decay = lr, beta_1, beta_2, epsilon, schedule_decay
        self.m = torch.zeros(nparams, dtype=torch.float32, device=device)
        self.v = torch.zeros(nparams, dtype=torch.float32, device=device)
        self.iterations = 0
        self.m_schedule = C.c_double(1.0)

    def step(self, params, grads, grad_scale=1.0):
        self.iterations += 1
        with torch.cuda.device(params.device):
            rc = self.lib.dj_nadam_step(_lib.ptr(params), _lib.ptr(grads), _lib.ptr(self.m), _lib.ptr(self.v),
                                        params.numel(), self.iterations, C.byref(self.m_schedule), self.lr,
                                        self.beta_1, self.beta_2, self.epsilon, self.schedule_decay,
                                        float(grad_scale), _stream_ptr())
        _lib.check(rc, "dj_nadam_step")
