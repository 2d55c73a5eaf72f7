"""TEST INFRASTRUCTURE: CPU oracle for the hot path (see oracle/ref_cpu.py header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
