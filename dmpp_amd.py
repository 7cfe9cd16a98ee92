"""Import shim: the package directory `decision-making-and-path-planning_amd/` is not a valid
Python identifier, so load it by path and re-export it as `dmpp_amd`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "decision-making-and-path-planning_amd")
_spec = importlib.util.spec_from_file_location("dmpp_amd_pkg", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["dmpp_amd_pkg"] = _mod
_spec.loader.exec_module(_mod)
globals().update({k: v for k, v in _mod.__dict__.items() if not k.startswith("__")})
