"""Import alias: the package directory is `mixgan-tts_amd/` (not a valid identifier), so
`import mixgan_tts_amd` resolves here and this module turns itself into that package."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "mixgan-tts_amd")]
__package__ = __name__
if __spec__ is not None:
    __spec__.submodule_search_locations = __path__
__file__ = _os.path.join(__path__[0], "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
