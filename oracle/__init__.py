"""CPU oracle for the NMF multiplicative-update hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in ``muscle_synergies_amd/`` (the product)
imports this package; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may, and there only as the checker.
"""
