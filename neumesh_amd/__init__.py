"""neumesh_amd -- MI355X-native (gfx950) implementation of NeuMesh's volumetric render inner loop.

Only the hot path named by BASELINE.json:north_star is implemented: the
host side mirrors the reference's own Python interface for that path, the device side is
hand-written HIP behind the C ABI declared in ``include/neumesh_hip.h``.  Importing the
package does not need a GPU; the first call that needs the HIP library loads it and fails
loudly if it is missing (there is no CPU fallback in the product path).
"""

__all__ = ["synthetic"]
