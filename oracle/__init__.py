"""CPU oracle for the m-mode map-making path.  TEST INFRASTRUCTURE ONLY.

A NumPy/SciPy restatement of the reference algorithm (radiocosmology/draco,
``draco/analysis/transform.py``, ``draco/analysis/mapmaker.py``,
``draco/synthesis/stream.py``), each function citing the reference lines it
follows.  It is the *checker* for the HIP path and the ``cpu_baseline`` leg of
``bench.py``; it is never the thing shipped or measured as the product:

* only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
  ``cpu_baseline`` leg import it;
* nothing under ``draco_amd/`` imports it (``tests/test_layout.py`` enforces this).

Pinning status (see DESIGN.md section "Oracle"):

* ``oracle.transform`` / ``oracle.mapmaker`` kernels are pinned against outputs of
  the reference's own functions executed in the build container
  (``oracle/gen_golden.py`` -> ``tests/golden/*.npz``).
* ``oracle.sht`` (HEALPix spherical-harmonic transforms) restates the published
  HEALPix/healpy convention; healpy/cora are absent here and the reference holds
  no fixture for it: **parity unpinned** for that stage, validated by
  mathematical identities only.
* The reference's own test-suite holds no golden vector for any function on
  this path.
"""
