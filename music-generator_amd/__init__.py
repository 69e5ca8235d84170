"""DeepJ (calclavia/music-generator) hot path for AMD MI355X (gfx950).

Biaxial-LSTM training step (teacher-forced forward + BPTT + Nadam) and the
autoregressive sampling step, as hand-written HIP kernels behind a C ABI
(include/deepj_hip.h), exposed through the reference's model.py / train.py /
generate.py surface.  `install()` registers those module names so reference-style
callers (`from model import build_models`) resolve to this implementation.
"""
__version__ = "0.1.0"

_DROPIN = ("constants", "util", "model", "dataset", "midi_util", "generate", "train")


def install():
    """Alias this package's modules under the reference's flat module names."""
    import importlib
    import sys
    for name in _DROPIN:
        try:
            sys.modules[name] = importlib.import_module(__name__ + "." + name)
        except ModuleNotFoundError as e:          # module of a later milestone
            if e.name != __name__ + "." + name:
                raise
