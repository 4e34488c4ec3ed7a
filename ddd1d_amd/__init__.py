"""Import alias for the package directory ``data-driven-discretization-1d_amd``.

The product directory carries the reference's repository name and therefore a
hyphenated, non-importable directory name.  This shim makes it importable as
``ddd1d_amd`` by pointing the package search path at that directory and
executing its ``__init__``; it contains no logic of its own.
"""
import os as _os

_HERE = _os.path.dirname(_os.path.abspath(__file__))
_REAL = _os.path.join(_os.path.dirname(_HERE), 'data-driven-discretization-1d_amd')
if not _os.path.isdir(_REAL):
  raise ImportError('package directory not found: ' + _REAL)
__path__.insert(0, _REAL)
with open(_os.path.join(_REAL, '__init__.py')) as _f:
  exec(compile(_f.read(), _os.path.join(_REAL, '__init__.py'), 'exec'))
del _f
