"""CPU oracle for the space-carving hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package; nothing under ``plant-3d-vision_amd/`` does.

Two independent restatements of the reference kernels
(``plant3dvision/kernels/backprojection.c``, ``common.h``):

* :mod:`oracle.oracle_c`  -- ctypes binding of ``spacecarve_oracle.c`` (strict IEEE C99),
* :mod:`oracle.oracle_np` -- vectorised NumPy float32 restatement.

Parity pin status is recorded in DESIGN.md ("Oracle").
"""
