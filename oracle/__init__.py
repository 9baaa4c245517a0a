"""ORACLE package: CPU restatements of the reference's hot-path algorithms.

Test infrastructure only.  Nothing under the product package imports this; only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg do (as the checker / the timed CPU baseline).
"""
