#!/usr/bin/env python3
"""Developer check: throughput of the CPU oracle (bench.py's cpu_baseline) at different BLAS thread counts."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
for n in (1, 4, 8, 16, 32, 128):
    sys.argv = ["bench.py", "--cpu-threads", str(n)]
    args = bench.parse()
    r = bench.cpu_baseline(args, 3.0)
    print(n, r["cores"], round(r["value"]))
