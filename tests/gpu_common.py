"""helpers for the -m gpu parity tests: build the device-side Field from the golden parameters"""
import numpy as np
import torch

from tests.common import load_golden, make_table


from avatarcraft_amd.synthetic import device_field            # noqa: E402,F401  (shared with bench.py / smoke())


def oracle_field(params, table):
    from oracle import oracle as O
    p = params
    return O.Field(table, p["offsets"], p["W1"], p["b1"], p["W2"], p["b2"], p["Wc1"], p["Wc2"], p["Wc3"], float(p["per_level_scale"]))


def bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def assert_bitwise(gpu, ref, name):
    g = gpu.detach().cpu().numpy() if isinstance(gpu, torch.Tensor) else np.asarray(gpu)
    r = np.asarray(ref)
    assert g.shape == r.shape, f"{name}: shape {g.shape} vs {r.shape}"
    if g.dtype.kind == "f":
        same = bits(g) == bits(r)
        # NaN payloads aside, every bit must match
        if not same.all():
            bad = np.argwhere(~same)
            i = tuple(bad[0])
            raise AssertionError(f"{name}: {len(bad)} of {g.size} values differ bitwise; first at {i}: gpu={g[i]!r} oracle={r[i]!r} "
                                 f"maxabs={np.nanmax(np.abs(g.astype(np.float64) - r.astype(np.float64)))}")
    else:
        assert np.array_equal(g, r), f"{name}: {int((g != r).sum())} integer mismatches"
