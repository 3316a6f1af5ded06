"""Randomised bit-exactness of point_to_mesh_distance's searches against the oracle (cases: tridist_fuzz_cases.py)."""
import pytest

from tridist_fuzz_cases import check_case


@pytest.mark.gpu
@pytest.mark.parametrize('first', [0, 40])
def test_triangle_distance_random_meshes_vs_oracle(first):
    failures = []
    for case in range(first, first + 40):
        desc, msgs = check_case(case)
        if msgs:
            failures.append(f'case {case} ({desc}): ' + '; '.join(msgs))
    assert not failures, '\n'.join(failures)
