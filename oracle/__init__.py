"""CPU oracle of the MFSR hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package, and only as the checker.  PARITY UNPINNED: see oracle_common.h.
"""
