"""helpers for the -m gpu parity tests: build the device-side Field from the golden parameters"""
import numpy as np
import torch

from tests.common import load_golden, make_table


def device_field(params=None, device="cuda:0", rough=False):
    from avatarcraft_amd import nsr_ops
    p = params if params is not None else load_golden("nsr_params.npz")
    if rough:
        table = make_table(int(p["offsets"][-1]), seed=int(p["table_seed"]) + 1, amp=0.5)
    else:
        table = make_table(int(p["offsets"][-1]), seed=int(p["table_seed"]), offsets=p["offsets"], level_amp=p["level_amp"])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)
    f = nsr_ops.Field(t(table), p["offsets"], float(p["per_level_scale"]), 16, t(p["W1"]), t(p["b1"]), t(p["W2"]), t(p["b2"]),
                      t(p["Wc1"]), t(p["Wc2"]), t(p["Wc3"]))
    return f, table


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
