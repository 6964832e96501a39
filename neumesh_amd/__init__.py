"""neumesh_amd -- MI355X-native (gfx950) implementation of NeuMesh's volumetric render inner loop.

Only the hot path named by BASELINE.json:north_star is implemented: the
host side mirrors the reference's own Python interface for that path (same class / function
names and argument meaning), the device side is hand-written HIP behind the C ABI declared in
``include/neumesh_hip.h``.  Importing the package needs neither a GPU nor the built library;
the first call that needs the HIP library loads it and fails loudly if it (or a GPU) is
missing -- there is no CPU fallback in the product path.

    from neumesh_amd import build_framework          # models/frameworks/__init__.py
    from neumesh_amd import MeshGrid, NeuMesh, SingleRenderer, volume_render
    from neumesh_amd import frnn                     # drop-in for `import frnn`
    from neumesh_amd import Trainer                  # models/trainer.py (2nd element of get_model's tuple)
    from neumesh_amd import TextureEditableNeuMesh   # editing/texture_neumesh/texture_neumesh.py
    from neumesh_amd import ray_casting              # models/ray_casting.py (surface_render, root finding, sphere tracing)
    from neumesh_amd import frames                   # render.py:183-184, 219-249 (uint8 images of a frame, on the device)
"""
from . import synthetic  # noqa: F401  (numpy only)


def __getattr__(name):  # lazy: torch is imported only when the model classes are touched
    import importlib
    table = {
        "MeshGrid": "mesh_grid", "MeshPrimitive": "mesh_grid", "NeuMesh": "neumesh", "SingleRenderer": "renderer",
        "volume_render": "renderer", "get_model": "framework", "build_framework": "framework", "Trainer": "trainer",
        "TextureEditableNeuMesh": "editing", "surface_render": "ray_casting",
    }
    if name in table:
        return getattr(importlib.import_module(f".{table[name]}", __name__), name)
    if name in ("frnn", "mesh_grid", "neumesh", "renderer", "framework", "ply", "sharded", "rays", "build", "_lib", "trainer", "editing",
                "ray_casting", "frames"):
        return importlib.import_module(f".{name}", __name__)
    raise AttributeError(name)
