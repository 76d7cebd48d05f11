"""CPU oracle for the SAC hot path -- TEST INFRASTRUCTURE ONLY.

Nothing under ``oracle/`` is part of the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it, and only as the checker / the timed CPU baseline, never as a compute
fallback.  The product path (``robosuite_benchmark_amd``) fails loudly when
the HIP library is missing.

Parity status (see DESIGN.md "Oracle"):

* ``replay_index_stream`` -- pinned bit-for-bit against NumPy's own legacy
  ``RandomState`` (the very generator the reference seeds at
  scripts/train.py:112 and rlkit's ``random_batch`` draws from).
* ``sac_step_torch`` -- "rlkit-equivalent restatement".  rlkit is not vendored
  in /root/reference and cannot be imported, and the reference holds no
  bit-level golden vector for a gradient step: **parity unpinned** at the
  bit/epsilon level.  It is pinned against every known answer the shipped
  ``progress.csv`` files hold for this path (KA1..KA7).
"""
