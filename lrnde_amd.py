"""Import shim: the package directory is named `localregneuralde.jl_amd` (with a dot), which
the import statement cannot spell.  `import lrnde_amd` loads it under the module name
`localregneuralde_jl_amd` and re-exports its public names."""
import importlib.util
import os
import sys

_NAME = "localregneuralde_jl_amd"
_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "localregneuralde.jl_amd")

if _NAME not in sys.modules:
    _spec = importlib.util.spec_from_file_location(_NAME, os.path.join(_DIR, "__init__.py"),
                                                   submodule_search_locations=[_DIR])
    _mod = importlib.util.module_from_spec(_spec)
    sys.modules[_NAME] = _mod
    try:
        _spec.loader.exec_module(_mod)
    except BaseException:
        del sys.modules[_NAME]
        raise
pkg = sys.modules[_NAME]
globals().update({k: v for k, v in vars(pkg).items() if not k.startswith("__")})
