"""-m gpu: the opt-in face-list search (AC_WARP_FLIST=1: accel_cells_kernel builds per-cell face lists, warp_samples_flist_kernel resolves the samples
that have one, the tile-walk kernel the rest as a fixup pass) must return the exhaustive kernel's bits like the default search does.  The switch is read
once per process, so the existing bit-for-bit tests are re-run in a child process with it set.  (Measured slower than the tile walk on the bench frame
-- profiles/r04_experiments.txt -- which is why it is off by default; it stays tested because it stays in the library.)"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_face_list_search_is_bit_identical_to_brute_force():
    env = dict(os.environ, AC_WARP_FLIST="1")
    r = subprocess.run([sys.executable, "-m", "pytest", "tests", "-m", "gpu", "-q", "-x", "-p", "no:cacheprovider", "-k",
                        "test_warp_accel_equals_brute_force or test_warped_render_bitwise_vs_oracle or test_warped_render_skip_masked_tiles"],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    tail = r.stdout[-1500:]
    assert r.returncode == 0 and " passed" in tail and "failed" not in tail, tail
