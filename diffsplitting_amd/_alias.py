"""Reference module paths (``model.networks``, ``data.tiling_manager``, ``core.logger`` ...) for the engine-backed
packages.

The reference's scripts and notebooks import top-level packages ``model``, ``data`` and ``core`` (split.py:1-20,
``import model as Model``).  ``diffsplitting_amd/compat`` holds three-line packages of those names; with that directory
on ``PYTHONPATH`` an ``import model.networks`` resolves to ``diffsplitting_amd.model.networks`` -- the same module
object, imported under its real name (the packages use relative imports and must not be executed a second time under
another name).
"""
import importlib
import importlib.abc
import importlib.machinery
import sys

_ROOTS = set()


class _AliasLoader(importlib.abc.Loader):
    def __init__(self, real):
        self.real = real

    def create_module(self, spec):
        return self.real

    def exec_module(self, module):
        pass


class _AliasFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        root = fullname.split(".", 1)[0]
        if root not in _ROOTS or "." not in fullname:
            return None
        try:
            real = importlib.import_module("diffsplitting_amd." + fullname)
        except ModuleNotFoundError as e:
            if e.name == "diffsplitting_amd." + fullname:
                return None          # the engine has no such module: let the normal machinery report it
            raise
        return importlib.machinery.ModuleSpec(fullname, _AliasLoader(real), is_package=hasattr(real, "__path__"))


_FINDER = _AliasFinder()


def install(root):
    """Make ``root`` (``model`` | ``data`` | ``core``) and everything below it an alias of ``diffsplitting_amd.<root>``."""
    real = importlib.import_module("diffsplitting_amd." + root)
    _ROOTS.add(root)
    if _FINDER not in sys.meta_path:
        sys.meta_path.insert(0, _FINDER)
    sys.modules[root] = real
    return real
