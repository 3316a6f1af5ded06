"""Randomised parity of the fused DIB-R operator against the oracle (cases: dibr_fuzz_cases.py)."""
import pytest

from dibr_fuzz_cases import check_case


@pytest.mark.gpu
@pytest.mark.parametrize('first', [0, 40, 80])
def test_fused_dibr_random_scenes_vs_oracle(first):
    failures = []
    for case in range(first, first + 40):
        desc, msgs = check_case(case)
        if msgs:
            failures.append(f'case {case} ({desc}): ' + '; '.join(msgs))
    assert not failures, '\n'.join(failures)
