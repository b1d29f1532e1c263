"""Shared helpers for the tests: seeded synthetic inputs that both the golden generator
(tests/golden/make_golden.py, run once against the reference) and the parity tests rebuild
bit-identically, so large arrays (the 49 MB hash table) never have to be committed."""
import os
import numpy as np

TABLE_SEED = 20260926
N_TABLE_DEFAULT = 6119857            # sum of the 16 level sizes of the default model
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def make_table(n_entries=N_TABLE_DEFAULT, level_dim=2, seed=TABLE_SEED, amp=0.5, offsets=None, level_amp=None):
    """embeddings ~ U(-amp, amp), float32, from the frozen numpy RandomState stream.
    With offsets + level_amp the amplitude is per level ("smooth" field: fine levels carry little
    energy, like a trained avatar; the SDF then has |grad| ~ 1 instead of ~5)."""
    rs = np.random.RandomState(seed)
    t = rs.uniform(-1.0, 1.0, size=(n_entries, level_dim))
    if level_amp is not None:
        a = np.repeat(np.asarray(level_amp, np.float64), np.diff(np.asarray(offsets, np.int64)))
        t = t * a[:, None]
    else:
        t = t * amp
    return t.astype(np.float32)


def smooth_level_amp(scale):
    """per-level table amplitude 3/scale_l (level 0: 0.2 ... level 15: 0.0015)"""
    return 3.0 / np.asarray(scale, np.float64)


def make_rays(h, w, dist=1.7, f=None, jitter_seed=None, yaw=0.35, pitch=-0.2):
    """Pinhole rays: camera on a sphere of radius `dist` (yaw/pitch in rad) looking at the origin.
    Returns (rays_o[h*w,3], rays_d[h*w,3]) float32, d normalised."""
    f = f if f is not None else 0.78125 * w
    jj, ii = np.meshgrid(np.arange(w, dtype=np.float64), np.arange(h, dtype=np.float64), indexing="xy")
    px = (jj + 0.5 - w / 2) / f
    py = -(ii + 0.5 - h / 2) / f
    if jitter_seed is not None:
        rs = np.random.RandomState(jitter_seed)
        px = px + rs.uniform(-0.3, 0.3, px.shape) / f
        py = py + rs.uniform(-0.3, 0.3, py.shape) / f
    d_cam = np.stack([px, py, -np.ones_like(px)], -1).reshape(-1, 3)
    cy, sy, cp, sp = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch)
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rx = np.array([[1, 0, 0], [0, cp, -sp], [0, sp, cp]])
    R = Ry @ Rx
    d = d_cam @ R.T
    d = d / np.linalg.norm(d, axis=1, keepdims=True)
    o = np.tile((R @ np.array([0, 0, dist]))[None], (d.shape[0], 1))
    o = o.astype(np.float32); d = d.astype(np.float32)
    # one axis-aligned ray exercises the (d + 1e-15) path of near_far_from_bound
    o[0] = [0, 0, dist]; d[0] = [0, 0, -1]
    return o, d


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


def oracle_field_from_golden(params=None):
    from oracle import oracle as O
    p = params if params is not None else load_golden("nsr_params.npz")
    table = make_table(int(p["offsets"][-1]), seed=int(p["table_seed"]), offsets=p["offsets"], level_amp=p["level_amp"])
    return O.Field(table, p["offsets"], p["W1"], p["b1"], p["W2"], p["b2"], p["Wc1"], p["Wc2"], p["Wc3"],
                   float(p["per_level_scale"]))
