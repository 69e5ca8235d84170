"""Convert the reference's weights-only Keras HDF5 checkpoint (out/model.h5, reference train.py:23) into the .npz
this build loads (music_generator_amd.model.Model.load_weights).  Needs h5py (not in the build image: run it where
the checkpoint was made).

    python tools/convert_keras_h5.py out/model.h5 out/model.npz

Tensor layouts are identical (Keras Dense / Conv1D / LSTM layouts, gate blocks i, f, c, o), so this is a rename by
music_generator_amd.model.keras_name_map; when the layer names in the file differ from a fresh-process run (Keras
auto-names depend on what else the process had built), tensors are matched by creation order within each shape."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def convert(h5, layout, name_map):
    """h5: mapping layer group -> mapping weight name -> array (an h5py.File, or a dict in tests); layout:
    [(tensor name, shape)] of this build.  Returns {tensor name: float32 array}."""
    root = h5["model_weights"] if "model_weights" in h5 else h5
    out, missing = {}, []
    for name, shape in layout:
        grp, wname = name_map[name]
        if grp in root and wname in root[grp]:
            w = np.asarray(root[grp][wname], np.float32)
            if w.shape != tuple(shape):
                raise ValueError("%s: file has %s, model expects %s" % (wname, w.shape, tuple(shape)))
            out[name] = w
        else:
            missing.append((name, tuple(shape)))
    if missing:
        # fall back: remaining file tensors in file order, matched to the missing ones by shape, first come first served
        used = {name_map[n][1] for n in out}
        pool = [(wn, np.asarray(root[g][wn], np.float32)) for g in root for wn in root[g] if wn not in used]
        for name, shape in missing:
            k = next((i for i, (_, w) in enumerate(pool) if w.shape == shape), None)
            if k is None:
                raise KeyError("no tensor of shape %s left in the file for %s" % (shape, name))
            out[name] = pool.pop(k)[1]
    return out


def main(argv):
    import h5py
    from music_generator_amd.engine import DeepJConfig
    from music_generator_amd.model import keras_name_map
    from param_layout import layout_of
    src, dst = argv
    cfg = DeepJConfig()
    with h5py.File(src, "r") as f:
        arrays = convert(f, layout_of(cfg), keras_name_map(cfg))
    np.savez(dst, **arrays)
    print("wrote", dst, "with", len(arrays), "tensors,", sum(a.size for a in arrays.values()), "parameters")


if __name__ == "__main__":
    main(sys.argv[1:3])
