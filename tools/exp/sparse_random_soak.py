"""The seeded random-configuration test of the sparse node (tests/test_gpu_parity.py test_sparse_backward_random_configurations) over many more
seeds than the suite runs -- a one-off confidence run, not a test.   python tools/exp/sparse_random_soak.py [first] [last]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import pytest
import test_gpu_parity as T
a, b = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (10, 70)
bad = 0
for seed in range(a, b):
    mp = pytest.MonkeyPatch()
    try:
        T.test_sparse_backward_random_configurations(seed, mp)
    except AssertionError as e:
        bad += 1
        print("FAILED seed", seed, str(e)[:300])
    finally:
        mp.undo()
print(f"seeds {a} .. {b - 1}: {b - a - bad} passed, {bad} failed")
