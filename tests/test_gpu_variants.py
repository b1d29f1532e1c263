"""-m gpu: compile-time alternatives of the product that must stay correct.  `avatarcraft_amd.build` links them next to the product
(`libavatarcraft_hip_<name>.so`); a test process loads one through AC_LIB_PATH and runs the parity tests that cover the changed code.

rec12: the table-gradient scatter with full-fp32 12-byte queue records (-DAC_REC8=0).  The shipped 8-byte records round every contribution to 16 / 17
mantissa bits before the fixed-point sum; the fp32 form is the reference's precision (fp32 atomicAdd) and has to keep passing the same fp64-oracle bound."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_full_fp32_scatter_records_pass_the_backward_parity_tests():
    from avatarcraft_amd.build import variant_path
    so = variant_path("rec12")
    assert os.path.exists(so), f"{so} is missing: python -m avatarcraft_amd.build links it next to the product"
    env = dict(os.environ, AC_LIB_PATH=so)
    sel = "test_hip_backward_matches_oracle_backward_on_the_4096_ray_patch or test_hash_forward_backward or test_sds_step_without_autograd_equals_autograd_step"
    r = subprocess.run([sys.executable, "-m", "pytest", "tests", "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider", "-k", sel], env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-3000:]
    assert " passed" in r.stdout and "failed" not in r.stdout.splitlines()[-1]
