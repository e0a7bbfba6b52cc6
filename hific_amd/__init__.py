"""Importable alias for the package directory `high-fidelity-generative-compression_amd/` (a hyphenated name cannot
be imported directly).  `import hific_amd` executes that directory's __init__ with this module as the package."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "high-fidelity-generative-compression_amd")
__path__ = [_real]
__file__ = _os.path.join(_real, "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
