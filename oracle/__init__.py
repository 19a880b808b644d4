"""CPU oracle for the embedding hot path -- test infrastructure, never part of the product path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
