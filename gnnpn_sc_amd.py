"""Import alias: the product package lives in the directory ``gnnpn-sc_amd/`` (a name Python
cannot import directly because of the hyphen).  ``import gnnpn_sc_amd`` loads that directory as
the package ``gnnpn_sc_amd`` and replaces this stub in ``sys.modules``."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gnnpn-sc_amd")
_spec = importlib.util.spec_from_file_location(
    "gnnpn_sc_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["gnnpn_sc_amd"] = _mod
_spec.loader.exec_module(_mod)
