"""Importable alias for the package directory `when-do-gnns-help_amd/` (a hyphen cannot appear in a Python
module name).  `import wdg_amd` == the package that lives in that directory."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "when-do-gnns-help_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
del _os, _f
