"""Seeded synthetic inputs of the benchmark and the parity tests (no external data exists offline: SURVEY 8c / 8d): the hash table, camera rays,
the SMPL-sized body with per-vertex transforms, and the field parameters (`data/synthetic_field.npz`: seed-0 geometric init of the reference's
NeRFNetwork with randomised feature columns, written by tests/golden/make_golden.py against the reference; a byte copy of tests/golden/nsr_params.npz).
bench.py, __graft_entry__.smoke() and tests/ all build their inputs here, so that the product benchmark does not import the test package."""
import os

import numpy as np

TABLE_SEED = 20260926
N_TABLE_DEFAULT = 6119857            # sum of the 16 level sizes of the default model
DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")


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


def make_body(n_lat=40, n_lon=80, seed=3):
    """Synthetic SMPL-like body for the warp tests: a closed capsule mesh (UV sphere stretched along y, scaled to the
    avatar's size), per-vertex rigid-ish 4x4 transforms (float64, smooth in space, like blended LBS matrices).
    Returns verts[V,3] f32, faces[F,3] i32, Ts[V,4,4] f64."""
    th = np.linspace(0, np.pi, n_lat + 2)[1:-1]
    ph = np.linspace(0, 2 * np.pi, n_lon, endpoint=False)
    v = [[0, 1, 0]]
    for t in th:
        for p_ in ph:
            v.append([np.sin(t) * np.cos(p_), np.cos(t), np.sin(t) * np.sin(p_)])
    v.append([0, -1, 0])
    v = np.array(v) * np.array([0.28, 0.85, 0.2])
    f = []
    for j in range(n_lon):
        f.append([0, 1 + (j + 1) % n_lon, 1 + j])
    for i in range(n_lat - 1):
        for j in range(n_lon):
            a = 1 + i * n_lon + j; b = 1 + i * n_lon + (j + 1) % n_lon; c = a + n_lon; d = b + n_lon
            f.append([a, b, d]); f.append([a, d, c])
    last = len(v) - 1
    for j in range(n_lon):
        a = 1 + (n_lat - 1) * n_lon + j; b = 1 + (n_lat - 1) * n_lon + (j + 1) % n_lon
        f.append([a, b, last])
    verts = v.astype(np.float32)
    faces = np.array(f, dtype=np.int32)
    rs = np.random.RandomState(seed)
    # smooth transform field: rotation about z by an angle that varies with height + small translation, scaled by 1/0.9
    ang = 0.35 * np.sin(2.0 * verts[:, 1].astype(np.float64)) + 0.05 * rs.normal(size=verts.shape[0])
    Ts = np.tile(np.eye(4)[None], (verts.shape[0], 1, 1))
    Ts[:, 0, 0] = np.cos(ang); Ts[:, 0, 1] = -np.sin(ang); Ts[:, 1, 0] = np.sin(ang); Ts[:, 1, 1] = np.cos(ang)
    Ts[:, :3, 3] = 0.03 * rs.normal(size=(verts.shape[0], 3))
    Ts = Ts @ (np.eye(4) / 0.9)
    return verts, faces, Ts


def make_body_sequence(n_frames=20, n_lat=83, n_lon=83, seed=7):
    """A synthetic ANIMATION of make_body's capsule for the posed-frame bench (BASELINE config 4 stands on an AMASS pose sequence, which is not in this
    image): the rest mesh bent by two smooth "joints" -- a rotation about z whose angle grows with height above the hip and one about x below it -- with
    angles that swing sinusoidally over the sequence (extremities move up to ~8 cm per frame, the size of a limb's motion between AMASS frames at 12 fps).
    Per frame: verts [V,3] f32 (the posed mesh) and Ts [V,4,4] f64 = bend(frame) @ make_body's rest transforms (rest -> scene, like blended LBS matrices).
    -> (list of verts, faces, list of Ts)"""
    verts0, faces, T0 = make_body(n_lat=n_lat, n_lon=n_lon)
    rs = np.random.RandomState(seed)
    ph = rs.uniform(0, 2 * np.pi, 2)
    y = verts0[:, 1].astype(np.float64)
    up, dn = np.clip(y / 0.85, 0.0, 1.0) ** 2, np.clip(-y / 0.85, 0.0, 1.0) ** 2          # smooth joint weights: 0 at the hip, 1 at the ends
    out_v, out_T = [], []
    for t in range(n_frames):
        a = 0.5 * np.sin(2 * np.pi * t / 20.0 + ph[0]) * up                              # about z, pivot at the origin
        b = 0.4 * np.sin(2 * np.pi * t / 20.0 + ph[1]) * dn                              # about x
        B = np.tile(np.eye(4)[None], (verts0.shape[0], 1, 1))
        ca, sa, cb, sb = np.cos(a), np.sin(a), np.cos(b), np.sin(b)
        Rz = np.zeros((verts0.shape[0], 3, 3)); Rz[:, 0, 0] = ca; Rz[:, 0, 1] = -sa; Rz[:, 1, 0] = sa; Rz[:, 1, 1] = ca; Rz[:, 2, 2] = 1
        Rx = np.zeros((verts0.shape[0], 3, 3)); Rx[:, 0, 0] = 1; Rx[:, 1, 1] = cb; Rx[:, 1, 2] = -sb; Rx[:, 2, 1] = sb; Rx[:, 2, 2] = cb
        B[:, :3, :3] = Rz @ Rx
        out_v.append(np.einsum("vij,vj->vi", B[:, :3, :3], verts0.astype(np.float64)).astype(np.float32))
        out_T.append(B @ T0)
    return out_v, faces, out_T


def load_field_params():
    """the synthetic field of BASELINE configs 2 - 5 (W1, b1, W2, b2, Wc1..3, offsets, level amplitudes, inv_s, state-dict entries)"""
    return dict(np.load(os.path.join(DATA, "synthetic_field.npz"), allow_pickle=False))


def field_table(p, rough=False):
    """the 49 MB hash table of that field, regrown from its seed (never committed)"""
    if rough:
        return make_table(int(p["offsets"][-1]), seed=int(p["table_seed"]) + 1, amp=0.5)
    return make_table(int(p["offsets"][-1]), seed=int(p["table_seed"]), offsets=p["offsets"], level_amp=p["level_amp"])


def device_field(params=None, device="cuda:0", rough=False):
    """(nsr_ops.Field on the device, host copy of the table)"""
    import torch
    from . import nsr_ops
    p = params if params is not None else load_field_params()
    table = field_table(p, rough)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(device)
    f = nsr_ops.Field(t(table), p["offsets"], float(p["per_level_scale"]), 16, t(p["W1"]), t(p["b1"]), t(p["W2"]), t(p["b2"]),
                      t(p["Wc1"]), t(p["Wc2"]), t(p["Wc3"]))
    return f, table
