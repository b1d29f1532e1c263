"""Shared helpers for the tests: seeded synthetic inputs that both the golden generator
(tests/golden/make_golden.py, run once against the reference) and the parity tests rebuild
bit-identically, so large arrays (the 49 MB hash table) never have to be committed."""
import os
import numpy as np

# the generators themselves live in the package (bench.py and smoke() use them too, and must not import the test package)
from avatarcraft_amd.synthetic import TABLE_SEED, N_TABLE_DEFAULT, make_table, smooth_level_amp, make_rays, make_body   # noqa: F401

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name):
    return dict(np.load(os.path.join(GOLDEN, name), allow_pickle=False))


def oracle_field_from_golden(params=None):
    from oracle import oracle as O
    p = params if params is not None else load_golden("nsr_params.npz")
    table = make_table(int(p["offsets"][-1]), seed=int(p["table_seed"]), offsets=p["offsets"], level_amp=p["level_amp"])
    return O.Field(table, p["offsets"], p["W1"], p["b1"], p["W2"], p["b2"], p["Wc1"], p["Wc2"], p["Wc3"],
                   float(p["per_level_scale"]))


def edge_case_rays():
    """rays the slab test and the samplers rarely see (tests/golden/run_edge_*.npz): parallel to an axis (a zero direction component:
    the reference divides by d + 1e-15), starting inside the cube, missing the cube (far < near: the coarse z run backwards and the
    first torch.sort of cat_z_vals really sorts), grazing a face, pointing away, an unnormalised direction"""
    ro = np.array([[0.0, 0.0, 1.44], [0.0, 0.0, 1.44], [0.1, -0.2, 0.3], [3.0, 3.0, 3.0], [3.0, 0.0, 0.0], [1.6, 0.2, 2.0],
                   [0.0, 0.0, 1.44], [0.3, 1.7, 0.1], [-2.5, 0.4, 0.2], [0.0, 0.0, 1.44], [0.2, 0.1, 2.2], [1.599, 1.599, 2.5]], np.float32)
    rd = np.array([[0.0, 0.0, -1.0], [0.0, 1.0, 0.0], [0.0, 0.0, -1.0], [1.0, 1.0, 1.0], [-1.0, 0.0, 0.0], [0.0, 0.0, -1.0],
                   [0.0, 0.0, 1.0], [0.0, -1.0, 0.0], [1.0, 1e-9, -1e-9], [1e-4, -1e-4, -1.0], [0.0, 0.0, -3.0], [0.0, 0.0, -1.0]], np.float32)
    nrm = np.linalg.norm(rd, axis=1, keepdims=True); nrm[10] = 1.0              # ray 10 keeps its unnormalised direction
    return ro, (rd / nrm).astype(np.float32)
