"""CPU oracle for the MOFO / VideoMAE pretraining hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``mofo_amd/`` may import this package;
only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg use it, and there only as the checker / the timed CPU baseline.
"""
