"""Import shim: the product package lives in the directory ``text-to-image_amd/`` (not a legal Python identifier);
``import t2i_amd`` loads it under that name."""
import importlib.util
import os
import sys

_pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'text-to-image_amd')
_spec = importlib.util.spec_from_file_location('t2i_amd', os.path.join(_pkg_dir, '__init__.py'),
                                               submodule_search_locations=[_pkg_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules['t2i_amd'] = _mod
_spec.loader.exec_module(_mod)
