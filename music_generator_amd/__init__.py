"""Import alias: the product package lives in ``music-generator_amd/`` (the name the
build contract fixes, not importable because of the hyphen); this shim makes it
importable as ``music_generator_amd`` by pointing the package path there."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "music-generator_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
