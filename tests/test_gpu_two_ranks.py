"""Two real ranks on one GPU (gloo process group over device tensors): the data-parallel train step of BASELINE config 4 as far
as a single-GPU box allows -- tests/two_rank_gloo_gpu_check.py states what is compared."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_two_rank_train_step_matches_averaged_tower_gradients():
    here = os.path.dirname(os.path.abspath(__file__))
    r = subprocess.run([sys.executable, os.path.join(here, 'two_rank_gloo_gpu_check.py')], capture_output=True, text=True,
                       timeout=900, cwd=os.path.dirname(here))
    assert 'TWO_RANK_OK' in r.stdout, (r.returncode, r.stdout[-3000:], r.stderr[-3000:])
