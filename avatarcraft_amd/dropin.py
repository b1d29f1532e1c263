"""Register the avatarcraft_amd packages under the reference's top-level module names, so that the
reference's drivers (stylize.py, render_canonical.py, render_warp.py) import the MI355X implementations
without any edit:  `import avatarcraft_amd.dropin as d; d.install()` before importing the driver."""
import importlib
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


def install(force=False):
    """sys.modules[reference name] = avatarcraft_amd module.  Refuses to shadow an already imported
    module of that name unless force=True."""
    for ref_name, ours in _MAP.items():
        if ref_name in sys.modules and not force and not sys.modules[ref_name].__name__.startswith("avatarcraft_amd"):
            raise RuntimeError(f"module {ref_name!r} is already imported from {getattr(sys.modules[ref_name], '__file__', '?')}")
        sys.modules[ref_name] = importlib.import_module(ours)
    return sorted(_MAP)


def uninstall():
    for ref_name in _MAP:
        m = sys.modules.get(ref_name)
        if m is not None and m.__name__.startswith("avatarcraft_amd"):
            del sys.modules[ref_name]
