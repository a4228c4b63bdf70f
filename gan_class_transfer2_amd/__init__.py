"""import alias for the hyphenated package directory `gan-class-transfer2_amd/` (not importable by name)."""
import os as _os

__path__.insert(0, _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "gan-class-transfer2_amd"))
_init = _os.path.join(__path__[0], "__init__.py")
with open(_init) as _f:
    exec(compile(_f.read(), _init, "exec"))
