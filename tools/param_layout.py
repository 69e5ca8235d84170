"""[(tensor name, shape)] of a configuration straight from the library (dj_param_info); no GPU needed."""


def layout_of(cfg):
    from music_generator_amd.engine import param_layout
    return [(n, s) for n, _, s in param_layout(cfg)]
