"""Import shim: the package directory is ``camouflaged-vlm_amd/`` (a hyphen is not importable),
so this module becomes that package under the importable name ``camouflaged_vlm_amd``."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "camouflaged-vlm_amd")]
__file__ = _os.path.join(__path__[0], "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
