"""Import alias: `legal_rag_amd` -> the product directory `legal-rag_amd/`."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "legal-rag_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py"), encoding="utf-8") as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
