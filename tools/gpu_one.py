#!/usr/bin/env python3
"""Dev tool: run ONE solve configuration a few times (for rocprofv3 --pmc passes): gpu_one.py MODEL N_SITES B [iters]."""
import pathlib, sys
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent))
import gpu_bench_dev as g
model, n, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
g.run(model, n, B, 'lrp12', 'auto', rtol=1e-6, atol=1e-8, iters=int(sys.argv[4]) if len(sys.argv) > 4 else 3)
