"""Import alias: the package directory is named ``moe-infinity_amd`` (hyphen, as the repo
layout prescribes), which Python cannot import by name.  This shim makes
``import moe_infinity_amd`` resolve to that directory."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "moe-infinity_amd")
__path__ = [_real]
__file__ = _os.path.join(_real, "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
