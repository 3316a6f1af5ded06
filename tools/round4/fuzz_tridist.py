"""Randomised bit-exactness sweep of point_to_mesh_distance's searches against the oracle (cases: tests/tridist_fuzz_cases.py).
usage (GPU box): python tools/round4/fuzz_tridist.py [n_cases] [first_seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from tridist_fuzz_cases import check_case

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad, t0 = 0, time.time()
for case in range(seed0, seed0 + n_cases):
    desc, msgs = check_case(case)
    if msgs:
        bad += 1
        print(f'case {case} FAILED ({desc}):', '; '.join(msgs), flush=True)
print(f'{n_cases} cases from seed {seed0}: {bad} failed, {time.time() - t0:.0f} s', flush=True)
sys.exit(1 if bad else 0)
