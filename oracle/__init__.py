"""CPU oracle — TEST INFRASTRUCTURE ONLY.

A numpy/pandas/scipy restatement of the reference algorithms on the hot path (Ouranosinc/xclim),
each function citing the reference file:line it follows.  It exists to check the HIP kernels and to time the
CPU baseline; nothing under ``xclim_amd/`` may import it.  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` use it.

Pinning:
* ``oracle.quantile`` / ``oracle.run_length`` kernels are checked against the reference's own pure-numpy / njit
  bodies executed verbatim in the build container (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``) and
  against the known answers of the reference's tests (``tests/test_utils.py:28-73`` ...).
* everything that goes through xarray in the reference (resample, rolling, where/shift/ffill) is restated by hand
  and pinned by the reference's synthetic known-answer tests, ported in ``tests/test_oracle_reference_answers.py``.
* ``oracle.sdba`` (xsdba >= 0.4.0, NOT in the reference tree): **parity unpinned** — a specified restatement
  (numpy sort + Hyndman-Fan type 7, scipy ``interp1d``) validated analytically like ``tests/test_xsdba.py:113-155``.

Array convention: numpy arrays with TIME ON AXIS 0, any trailing dims; fp32 in unless stated.
"""
