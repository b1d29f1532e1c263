"""The oracle-vs-reference pins (golden vectors, known answers, host-mirror goldens) once more under the `gpu` marker.

They are CPU code and take seconds; the driver's round-end record is `pytest -m gpu` on the MI355X box, so running them there puts
the pins next to the GPU == oracle parity tests in the same record: oracle pinned to the reference, GPU pinned to the oracle.
(`-m "not gpu"` still runs the originals in their own modules.)"""
import pytest

from tests import test_host as _H
from tests import test_oracle_golden as _G
from tests import test_oracle_kat as _K

pytestmark = pytest.mark.gpu

for _m in (_G, _K):
    for _n in dir(_m):
        if _n.startswith("test_"):
            globals()[f"{_n}__pin"] = getattr(_m, _n)
for _n in ("test_encoder_surface_matches_reference_facts", "test_smpl_lbs_matches_reference_golden", "test_calc_local_trans_matches_reference_golden",
           "test_vanilla_nerf_plumbing_matches_reference", "test_style_paths_match_reference", "test_dataset_camera_rays_match_reference",
           "test_convert_amass_matches_reference_script", "test_sds_guidance_matches_reference_over_stub_models"):
    if hasattr(_H, _n):
        globals()[f"{_n}__pin"] = getattr(_H, _n)
del _m, _n
