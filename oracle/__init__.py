"""CPU oracle for the DeepJ hot path -- TEST INFRASTRUCTURE ONLY.

This package is a CPU restatement (torch-CPU / numpy, fp32 or fp64) of the
reference's biaxial-LSTM training step and sampling step
(/root/reference/model.py:14-169, train.py:18-29, generate.py:13-121).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import it, and only as the checker / the timed CPU baseline.  The
product package (``music-generator_amd``) never imports it and fails loudly when
its HIP library is missing.

PARITY STATUS: the Keras/TensorFlow arithmetic (LSTM cell, Conv1D 'same',
binary_crossentropy, Nadam) lives in un-vendored, un-pinned third-party
packages that are absent from /root/reference and from this image, and the
reference holds no test or fixture for model.py -> **parity unpinned** for that
arithmetic.  The non-Keras parts (sampling harness, dataset
windowing, apply_temperature, MIDI codec) ARE pinned against golden vectors
captured from the reference's own code (tests/golden/, make_golden.py).
"""
