"""Importable alias for the package directory `vcr-gaus_amd/` (a hyphen cannot be imported).

All sources live in `vcr-gaus_amd/`; this shim only redirects the package search path so that
`import vcr_gaus_amd.rasterizer` resolves to `vcr-gaus_amd/rasterizer.py`.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "vcr-gaus_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
