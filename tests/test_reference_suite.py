"""The drop-in claim, where the driver's GPU run sees it: the reference's OWN test files for the hot path, unchanged,
against kaolin_amd -- once through ``install_as_kaolin()`` and once with the reference's own Python layer (its autograd Functions)
bound to ``kaolin._C := kaolin_amd._C`` (KAMD_REF_LAYER=1).  The files are staged by ``tools/stage_reference_tests.sh`` into the
untracked scratch directory ``_ref_tests/`` (nothing of the reference is committed; the directory travels to the GPU box with the
tree); without it the tests skip.  Reference fixtures exercised: tests/python/kaolin/render/mesh/test_dibr.py:41-529,
test_rasterization.py:137-289, tests/python/kaolin/metrics/test_pointcloud.py:104-345, test_trianglemesh.py:26-152 (and
test_render, test_deftet, test_utils, test_check_sign, conversions/test_trianglemesh)."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STAGED = os.path.join(ROOT, '_ref_tests')
MIN_PASSED = 1192   # (profiles/r05final_reference_tests_and_fuzz.txt: 1192 passed, 377 skipped -- CPU / nvdiffrast / other-backend cases)


@pytest.mark.parametrize('ref_layer', [False, True], ids=['install_as_kaolin', 'reference_python_layer_over_C'])
def test_reference_tests_unchanged(ref_layer):
    if not os.path.isdir(os.path.join(STAGED, 'tests', 'python', 'kaolin')):
        pytest.skip('_ref_tests/ is not staged (bash tools/stage_reference_tests.sh needs /root/reference)')
    env = dict(os.environ)
    env.pop('KAMD_REF_LAYER', None)
    if ref_layer:
        env['KAMD_REF_LAYER'] = '1'
    res = subprocess.run([sys.executable, '-m', 'pytest', 'tests/python/kaolin', '-q', '-p', 'no:cacheprovider', '--import-mode=importlib'],
                         cwd=STAGED, env=env, capture_output=True, text=True, timeout=1500)
    tail = '\n'.join(res.stdout.splitlines()[-15:])
    summary = res.stdout.strip().splitlines()[-1] if res.stdout.strip() else ''
    passed = re.search(r'(\d+) passed', summary)
    failed = re.search(r'(\d+) (failed|error)', summary)
    assert res.returncode == 0 and failed is None, tail + '\n' + res.stderr[-2000:]
    assert passed is not None and int(passed.group(1)) >= MIN_PASSED, tail
