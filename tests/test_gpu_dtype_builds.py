"""Both builds of the library are products: libtcvom_hip.so (bf16 storage -- the north star's type, the default since round 5) and
libtcvom_hip_f16.so (fp16 storage, TCVOM_DTYPE=fp16 -- the type BASELINE config 5 names).  The suite runs in whichever type the
environment selects; this test runs the kernel-level tests, every whole-window test of the GCA+TAM path (reference goldens, benchmark-size
forward AND backward parity, gradient fidelity), the FBA+TAM tests (goldens, 544 x 960 oracle parity, the 1080p window) and the gradient
exchange tests in the OTHER type in a subprocess, so that one `pytest -m gpu` covers both builds with nothing excluded."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_the_other_storage_type_passes_its_parity_tests():
    from tcvom_amd._lib import DTYPE_NAME
    if os.environ.get('TCVOM_DTYPE_SUBTEST'):
        pytest.skip('already inside the other-dtype run')
    other = 'bf16' if DTYPE_NAME == 'fp16' else 'fp16'
    env = dict(os.environ, TCVOM_DTYPE=other, TCVOM_DTYPE_SUBTEST='1')
    cmd = [sys.executable, '-m', 'pytest', '-q', '-x', '-m', 'gpu', '-p', 'no:cacheprovider'] + \
          [os.path.join(REPO, 'tests', f) for f in ('test_gpu_ops.py', 'test_gpu_window.py', 'test_gpu_fba.py', 'test_gpu_ddp.py')]
    out = subprocess.run(cmd, cwd=REPO, env=env, capture_output=True, text=True, timeout=1800)
    tail = '\n'.join(out.stdout.splitlines()[-15:])
    print('TCVOM_DTYPE=%s:\n%s' % (other, tail))
    assert out.returncode == 0, tail + '\n' + out.stderr[-2000:]
    assert ' passed' in tail and ' failed' not in tail
