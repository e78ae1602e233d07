"""Import shim: the package directory is literally `abstractgps.jl_amd/` (not a valid dotted module
name), so load it under the module name `abstractgps_jl_amd`."""
import importlib.util
import sys
from pathlib import Path

_pkg_dir = Path(__file__).resolve().parent / "abstractgps.jl_amd"
_spec = importlib.util.spec_from_file_location(__name__, _pkg_dir / "__init__.py",
                                               submodule_search_locations=[str(_pkg_dir)])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
