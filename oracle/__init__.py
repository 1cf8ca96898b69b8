"""ORACLE -- test infrastructure only (CPU restatement of the reference path).

Importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg; the product package never imports it.
"""
