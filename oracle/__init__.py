"""oracle/ -- CPU checker for the HIP path.  TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this package; the product never does."""
