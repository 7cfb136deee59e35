"""Import alias: ``import cmdgen_amd`` loads the package directory ``cmd-gen_amd/``.

The package directory carries the project's name (with a hyphen, which Python
cannot import directly); this one-file shim registers it under an importable
name.  Nothing else lives here.
"""
import importlib.util as _ilu
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), 'cmd-gen_amd')
_spec = _ilu.spec_from_file_location('cmdgen_amd', _os.path.join(_dir, '__init__.py'),
                                     submodule_search_locations=[_dir])
_mod = _ilu.module_from_spec(_spec)
_sys.modules['cmdgen_amd'] = _mod
_spec.loader.exec_module(_mod)
