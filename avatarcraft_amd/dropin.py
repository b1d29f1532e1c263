"""Register the avatarcraft_amd packages under the reference's top-level module names, so that the
reference's drivers (stylize.py, render_canonical.py, render_warp.py) import the MI355X implementations
without any edit:  `import avatarcraft_amd.dropin as d; d.install()` before importing the driver."""
import importlib
import importlib.abc
import importlib.machinery
import sys

_MAP = {
    "encoder": "avatarcraft_amd.encoder",
    "encoder.freq_encoder": "avatarcraft_amd.encoder.freq_encoder",
    "encoder.hashencoder": "avatarcraft_amd.encoder.hashencoder",
    "encoder.hashencoder.hashgrid": "avatarcraft_amd.encoder.hashencoder.hashgrid",
    "encoder.hashencoder.backend": "avatarcraft_amd.encoder.hashencoder.backend",
    "encoder.shencoder": "avatarcraft_amd.encoder.shencoder",
    "encoder.shencoder.sphere_harmonics": "avatarcraft_amd.encoder.shencoder.sphere_harmonics",
    "encoder.shencoder.backend": "avatarcraft_amd.encoder.shencoder.backend",
    "raymarching": "avatarcraft_amd.raymarching",
    "raymarching.raymarching": "avatarcraft_amd.raymarching.raymarching",
    "raymarching.backend": "avatarcraft_amd.raymarching.backend",
}


# fused=True: single MODULES inside the reference's own packages are served from here while the rest of those packages stays the reference's --
# `import models.instant_nsr` / `from models import instant_nsr` (render_canonical.py:30, render_warp.py:21, stylize.py:17, reconstruct.py:14) get
# avatarcraft_amd.instant_nsr, i.e. NeRFNetwork().render() is the FUSED renderer (one launch per ray batch) with zero edits to the reference, instead of the
# reference's run() over the stand-alone encoder operators (~300 torch launches per batch).  models.smpl, models.diffusion, utils.* remain the reference's.
_FUSED_MAP = {
    "models.instant_nsr": "avatarcraft_amd.instant_nsr",
}


class _Redirect(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """serves the names of _FUSED_MAP; going through the import system (not just sys.modules) makes `from models import instant_nsr` bind the attribute on the
    reference's `models` package like any submodule import does"""

    def find_spec(self, fullname, path=None, target=None):
        if fullname in _FUSED_MAP:
            return importlib.machinery.ModuleSpec(fullname, self)
        return None

    def create_module(self, spec):
        return importlib.import_module(_FUSED_MAP[spec.name])

    def exec_module(self, module):
        pass


_redirect = _Redirect()


def install(force=False, fused=False):
    """sys.modules[reference name] = avatarcraft_amd module.  Refuses to shadow an already imported
    module of that name unless force=True.  fused=True additionally routes `models.instant_nsr` to this package's NeRFNetwork (the fused renderer)."""
    for ref_name, ours in _MAP.items():
        if ref_name in sys.modules and not force and not sys.modules[ref_name].__name__.startswith("avatarcraft_amd"):
            raise RuntimeError(f"module {ref_name!r} is already imported from {getattr(sys.modules[ref_name], '__file__', '?')}")
        sys.modules[ref_name] = importlib.import_module(ours)
    if fused:
        for ref_name in _FUSED_MAP:
            if ref_name in sys.modules and not force and not sys.modules[ref_name].__name__.startswith("avatarcraft_amd"):
                raise RuntimeError(f"module {ref_name!r} is already imported from {getattr(sys.modules[ref_name], '__file__', '?')}")
            sys.modules.pop(ref_name, None)
        if _redirect not in sys.meta_path:
            sys.meta_path.insert(0, _redirect)
    return sorted(_MAP) + (sorted(_FUSED_MAP) if fused else [])


def uninstall():
    for ref_name in list(_MAP) + list(_FUSED_MAP):
        m = sys.modules.get(ref_name)
        if m is not None and m.__name__.startswith("avatarcraft_amd"):
            del sys.modules[ref_name]
            parent, _, leaf = ref_name.rpartition(".")
            pm = sys.modules.get(parent)
            if pm is not None and getattr(pm, leaf, None) is m and not pm.__name__.startswith("avatarcraft_amd"):
                delattr(pm, leaf)
    if _redirect in sys.meta_path:
        sys.meta_path.remove(_redirect)
