"""CPU oracle for the gated-GCN hot path: test infrastructure, not product code.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this."""
