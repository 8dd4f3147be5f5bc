"""Importable alias of the package directory ``two-for-one-diffusion_amd`` (not an identifier).

``import dff_amd`` and ``import dff_amd.langevin`` resolve to the very same module objects as
``importlib.import_module("two-for-one-diffusion_amd[.langevin]")``.
"""
import importlib
import importlib.abc
import importlib.util
import os
import sys

_REAL = "two-for-one-diffusion_amd"
_ALIAS = __name__
_root = os.path.dirname(os.path.abspath(__file__))
if _root not in sys.path:
    sys.path.insert(0, _root)


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.startswith(_ALIAS + "."):
            return importlib.util.spec_from_loader(fullname, self)
        return None

    def create_module(self, spec):
        return importlib.import_module(_REAL + spec.name[len(_ALIAS):])

    def exec_module(self, module):
        pass


sys.meta_path.insert(0, _AliasFinder())
_pkg = importlib.import_module(_REAL)
sys.modules[_ALIAS] = _pkg
