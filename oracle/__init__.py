"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the vit_core hot path.

Nothing under ``oracle/`` is product code.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import it, and only as the checker (never as the thing measured or shipped).
The product path (``vit-ssl_amd/``) never imports this package and fails loudly
when the HIP library is missing.

Parity status: the reference's own tests hold no values for this path
(SURVEY.md section 8c), so the oracle is pinned by fixtures generated from the
reference itself (``tests/golden/make_golden.py`` imports
``/root/reference/vit_core`` in the build container; outputs committed under
``tests/golden/*.npz``) and checked in ``tests/test_oracle_golden.py``.
"""
