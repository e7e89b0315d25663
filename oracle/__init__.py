"""TEST INFRASTRUCTURE ONLY.  CPU restatements of the reference's arithmetic for the hot path, used
as the parity checker by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing
under espresso_amd/ imports this package."""
