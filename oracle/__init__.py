"""ORACLE package: CPU restatement of the reference's adaptation hot path.

Test infrastructure only.  Importers allowed: ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py``.  Nothing under ``onda_amd/`` imports it.
Parity of every module here is pinned by golden vectors captured from the imported
reference (``tests/golden/make_golden.py``; fixtures ``tests/golden/*.npz``).
"""
