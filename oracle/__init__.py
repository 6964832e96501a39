"""oracle/ -- CPU parity oracle for the NeuMesh volumetric-render hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``neumesh_amd/`` (the product) may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg do, and only as the checker / the timed CPU baseline.

Contents
--------
* ``knn_ref.c`` / ``knn.py``   -- declared-arithmetic exact K-NN (stands in for the external,
  un-vendored, un-pinned FRNN CUDA package the reference calls at
  ``models/mesh_grid.py:64-74,109-119``).  **Parity unpinned**: the reference ships no
  golden vectors for that boundary, so the arithmetic is declared in ``knn_ref.c``.
* ``field.py``   -- numpy fp32 restatement of ``models/mesh_grid.py:88-144``,
  ``models/base.py:52-70`` and ``models/frameworks/neumesh/neumesh.py:113-273``.
* ``render.py``  -- numpy fp32 restatement of ``models/renderer.py:13-368`` and
  ``utils/rend_util.py:179-199,276-319``.
* ``refimport/`` -- throw-away stub modules that let the *real* reference be imported in the
  build container (it is 100 % Python) so that ``gen_golden.py`` can (a) check this
  restatement against the reference's own code and (b) write the fixtures in
  ``tests/golden/``.  ``/root/reference`` does not exist on the GPU box; nothing at test /
  bench run time reads it.
"""
