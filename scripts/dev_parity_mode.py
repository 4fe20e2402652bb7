"""Developer: parity of a debug-selected kernel variant (CRAG_DEBUG_MODE) against the oracle."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from dev_parity import run
ok = True
for args in [(1000, 3, 10), (257, 1, 5), (31, 2, 10), (4096, 32, 10), (4096, 64, 10), (20000, 33, 10), (50000, 100, 32),
             (100000, 64, 10), (777, 65, 5), (100000, 32, 10)]:
    ok &= run(*args)
ok &= run(5000, 40, 10, mask_frac=0.1)
ok &= run(5000, 8, 10, mask_frac=0.001)
ok &= run(3000, 64, 20, dup=True)
print("MODE", os.environ.get("CRAG_DEBUG_MODE"), "ALL OK" if ok else "FAILURES")
