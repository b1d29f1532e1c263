"""SMPL linear blend skinning and the per-frame rest->scene transforms: counterpart of
models/smpl.py:107-161,249-300,351-446,549-655 and render_warp.py:127-222.

Per frame and small (24 joints, 6890 vertices): it stays host-side torch (fp32, like the reference) and produces the inputs of the
warp kernel (csrc/warp.hip): Ts[6914,4,4] fp64 and world_verts[6890,3] fp32.  Behaviour kept from the reference, including:
  * rodrigues: angle = |r + 1e-8| (the epsilon is added to the VECTOR before the norm, models/smpl.py:566);
  * the pose-corrective blend shapes are evaluated but NOT applied (v_posed = v_shaped, models/smpl.py:423): the skinned mesh is the
    shaped template moved by the blended joint transforms only;
  * with return_T the returned vertices are the UNPOSED shaped template (+ rest joints when concat_joints), models/smpl.py:432-436.
The licensed SMPL_NEUTRAL.pkl is not shipped: BodyModel takes the arrays, `from_pickle` reads the file when the user has it, and
`synthetic()` builds a body with the SMPL topology sizes (V=6890, J=24, 10 betas, 207 pose-basis rows) for tests and benchmarks."""
import os
import pickle

import numpy as np
import torch

SMPL_SCALE = 0.9                          # utils/constant.py:39
SMPL_PARENTS = (-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 20, 21)


def batch_rodrigues(rot_vecs, epsilon=1e-8, dtype=torch.float32):
    """axis-angle [N,3] -> rotation matrices [N,3,3]:  R = I + sin(a) K + (1 - cos(a)) K^2"""
    angle = torch.norm(rot_vecs + 1e-8, dim=1, keepdim=True)
    axis = rot_vecs / angle
    x, y, z = axis[:, 0], axis[:, 1], axis[:, 2]
    o = torch.zeros_like(x)
    K = torch.stack([o, -z, y, z, o, -x, -y, x, o], dim=1).view(-1, 3, 3)
    s = torch.sin(angle)[:, :, None]
    c = torch.cos(angle)[:, :, None]
    eye = torch.eye(3, dtype=dtype, device=rot_vecs.device)[None]
    return eye + s * K + (1 - c) * torch.bmm(K, K)


def transform_mat(R, t):
    """[B,3,3], [B,3,1] -> [B,4,4] = [[R t],[0 1]]"""
    B = R.shape[0]
    M = torch.zeros((B, 4, 4), dtype=R.dtype, device=R.device)
    M[:, :3, :3] = R
    M[:, :3, 3:] = t
    M[:, 3, 3] = 1
    return M


def batch_rigid_transform(rot_mats, joints, parents, dtype=torch.float32):
    """kinematic chain: world transform of every joint and the transform relative to the rest joint position
    (A_j = G_j with G_j * [J_j;0] removed from the translation column)"""
    B, N = joints.shape[:2]
    parents = [int(p) for p in parents]
    rel = joints.clone()
    rel[:, 1:] = joints[:, 1:] - joints[:, parents[1:]]
    local = transform_mat(rot_mats.reshape(-1, 3, 3), rel.reshape(-1, 3, 1)).view(B, N, 4, 4)
    chain = [local[:, 0]]
    for j in range(1, N):
        chain.append(torch.matmul(chain[parents[j]], local[:, j]))
    G = torch.stack(chain, dim=1)
    posed_joints = G[:, :, :3, 3]
    jh = torch.cat([joints, torch.zeros_like(joints[..., :1])], dim=-1)[..., None]        # [B,N,4,1]
    corr = torch.matmul(G, jh)                                                               # [B,N,4,1]
    A = G - torch.cat([torch.zeros((B, N, 4, 3), dtype=G.dtype, device=G.device), corr], dim=-1)
    return posed_joints, A


def blend_shapes(betas, shape_disps):
    return torch.einsum('bl,mkl->bmk', [betas, shape_disps])


def vertices2joints(J_regressor, vertices):
    return torch.einsum('bik,ji->bjk', [vertices, J_regressor])


def lbs(betas, pose, v_template, shapedirs, posedirs, J_regressor, parents, lbs_weights, pose2rot=True, dtype=torch.float32,
        return_T=False, concat_joints=False):
    """same signature and returns as models/smpl.py:351; see the module docstring for the kept quirks"""
    B = max(betas.shape[0], pose.shape[0])
    v_delta = blend_shapes(betas, shapedirs)
    v_shaped = v_template + v_delta
    J = vertices2joints(J_regressor, v_shaped)
    if pose2rot:
        rot_mats = batch_rodrigues(pose.view(-1, 3), dtype=dtype).view(B, -1, 3, 3)
    else:
        rot_mats = pose.view(B, -1, 3, 3)
    v_posed = v_shaped                   # pose-corrective offsets are not applied by the reference
    J_transformed, A = batch_rigid_transform(rot_mats, J, parents, dtype=dtype)
    nj = J_regressor.shape[0]
    W = lbs_weights.unsqueeze(0).expand(B, -1, -1)
    T = torch.matmul(W, A.view(B, nj, 16)).view(B, -1, 4, 4)
    if return_T:
        if concat_joints:
            return torch.cat([T, A], dim=1), torch.cat([v_posed, J], dim=1), v_delta
        return T, v_posed, v_delta
    vh = torch.cat([v_posed, torch.ones_like(v_posed[..., :1])], dim=2)
    verts = torch.matmul(T, vh.unsqueeze(-1))[:, :, :3, 0]
    return verts, J_transformed


def _t(a, dtype=torch.float32, device=None):
    if isinstance(a, torch.Tensor):
        return a.to(device=device, dtype=dtype)
    if 'scipy.sparse' in str(type(a)):
        a = a.todense()
    return torch.tensor(np.asarray(a), dtype=dtype, device=device)


class BodyModel:
    """the buffers of models/smpl.py:SMPL (:77-105) and its three entry points"""

    def __init__(self, v_template, shapedirs, posedirs, J_regressor, parents, lbs_weights, faces, device=None):
        self.device = device or torch.device('cpu')
        self.dtype = torch.float32
        self.v_template = _t(v_template, device=self.device)                 # [V,3]
        self.shapedirs = _t(shapedirs, device=self.device)                   # [V,3,10]
        posedirs = np.asarray(posedirs, dtype=np.float32) if not isinstance(posedirs, torch.Tensor) else posedirs
        if posedirs.ndim == 3:                                               # pickle layout [V,3,207] -> [207, V*3]
            posedirs = posedirs.reshape(-1, posedirs.shape[-1]).T
        self.posedirs = _t(posedirs, device=self.device)
        self.J_regressor = _t(J_regressor, device=self.device)               # [J,V]
        p = torch.as_tensor(np.asarray(parents, dtype=np.int64)).clone()
        p[0] = -1
        self.parents = p
        self.lbs_weights = _t(lbs_weights, device=self.device)               # [V,J]
        self.faces = np.asarray(faces)

    @classmethod
    def from_pickle(cls, model_path, gender='neutral', device=None):
        path = os.path.join(model_path, f'SMPL_{gender.upper()}.pkl') if os.path.isdir(model_path) else model_path
        if not os.path.exists(path):
            raise FileNotFoundError(f'Path {path} does not exist!')
        with open(path, 'rb') as f:
            d = pickle.load(f, encoding='latin1')
        return cls(d['v_template'], d['shapedirs'], d['posedirs'], d['J_regressor'], d['kintree_table'][0], d['weights'], d['f'], device)

    @classmethod
    def synthetic(cls, seed=0, n_verts=6890, n_joints=24, faces=None, v_template=None):
        """random but well-formed buffers with the SMPL sizes: convex skinning weights over 4 joints per vertex, a sparse joint
        regressor with rows summing to 1, small shape / pose bases"""
        g = np.random.default_rng(seed)
        V, J = n_verts, n_joints
        if v_template is None:
            v_template = (g.standard_normal((V, 3)) * np.array([0.25, 0.55, 0.12])).astype(np.float32)
        shapedirs = (g.standard_normal((V, 3, 10)) * 0.01).astype(np.float32)
        posedirs = (g.standard_normal((V, 3, (J - 1) * 9)) * 0.005).astype(np.float32)
        Jr = np.zeros((J, V), np.float32)
        for j in range(J):
            idx = g.choice(V, 32, replace=False)
            w = g.random(32).astype(np.float32)
            Jr[j, idx] = w / w.sum()
        W = np.zeros((V, J), np.float32)
        for k in range(4):
            W[np.arange(V), g.integers(0, J, V)] += g.random(V).astype(np.float32) + 0.05
        W /= W.sum(1, keepdims=True)
        if faces is None:
            faces = g.integers(0, V, (13776, 3))
        return cls(v_template, shapedirs, posedirs, Jr, np.array(SMPL_PARENTS[:J]), W, faces)

    def to(self, device):
        """the same body with its buffers on `device` (lbs then runs there; calc_local_trans still hands numpy to the warp's uploader)"""
        device = torch.device(device)
        b = object.__new__(BodyModel)
        b.__dict__.update(self.__dict__)
        b.device = device
        for k in ("v_template", "shapedirs", "posedirs", "J_regressor", "lbs_weights"):
            setattr(b, k, getattr(self, k).to(device))
        return b

    def _args(self):
        return (self.v_template, self.shapedirs, self.posedirs, self.J_regressor, self.parents, self.lbs_weights)

    def verts_transformations(self, poses, betas, transl=None, return_tensor=True, concat_joints=False):
        """-> (vertices, T, delta_v): the per-vertex 4x4 (T-pose -> posed), models/smpl.py:107-161"""
        assert poses.shape[0] == 1
        poses, betas = _t(poses, self.dtype, self.device), _t(betas, self.dtype, self.device)
        L, vertices, delta_v = lbs(betas, poses, *self._args(), dtype=self.dtype, return_T=True, concat_joints=concat_joints)
        if transl is not None:
            M = torch.eye(4, dtype=self.dtype, device=self.device)[None].clone()
            M[0, :3, 3] = _t(transl, self.dtype, self.device).reshape(3)
            T = torch.matmul(M, L)
        else:
            T = L
        if not return_tensor:
            vertices = vertices.detach().cpu().numpy()[0]
            T = T.detach().cpu().numpy()[0]
        return vertices, T, delta_v

    def forward(self, poses, betas, transl=None, return_joints=False, return_tensor=True):
        """posed vertices (and joints), models/smpl.py:249-300"""
        assert poses.shape[0] == 1
        poses, betas = _t(poses, self.dtype, self.device), _t(betas, self.dtype, self.device)
        vertices, joints = lbs(betas, poses, *self._args(), dtype=self.dtype)
        if transl is not None:
            tr = _t(transl, self.dtype, self.device)
            vertices = vertices + tr.unsqueeze(1)
            joints = joints + tr.unsqueeze(1)
        if not return_tensor:
            vertices, joints = vertices.detach().cpu().numpy(), joints.detach().cpu().numpy()
        return (vertices[0], joints[0]) if return_joints else vertices[0]

    __call__ = forward


def convert_amass(npz, sample_rate=10):
    """AMASS sequence -> the pose array render_warp.py reads with --poseseq_path (utils/convert_amass.py:1-19): the first 63 axis-angle
    values of every `sample_rate`-th frame (root + 20 body joints), the 3 x 3 hand values zeroed, -> [F, 24, 3] float32.  `npz` is a path
    or an opened np.load(...) mapping with "poses" [F, >= 63]; returns (poses, betas[:10]) -- the script computes betas too but never stores them."""
    data = np.load(npz) if isinstance(npz, (str, os.PathLike)) else npz
    poses = np.asarray(data["poses"])[:, :63][::sample_rate]
    betas = np.asarray(data["betas"])[:10] if "betas" in data else None
    poses = np.concatenate([poses, np.zeros((poses.shape[0], 9), poses.dtype)], axis=1).reshape(-1, 24, 3).astype(np.float32)
    return poses, betas


def save_pose_sequence(path, poses):
    """the file format of utils/convert_amass.py:16-17 / render_warp.py:28-30: a raw np.save stream (whatever the extension)"""
    with open(path, "wb") as f:
        np.save(f, np.asarray(poses, np.float32))


def load_pose_sequence(path):
    """render_warp.py:28-30; calc_local_trans indexes poses[i][None], so [F,24,3] and [F,72] both work -- flattened here to [F,72]"""
    with open(path, "rb") as f:
        return np.load(f).astype(np.float32).reshape(-1, 72)


def da_pose():
    """the canonical "da" pose of NeuMan: legs spread by +-1 rad about z (render_warp.py:164-169)"""
    p = np.zeros((24, 3), np.float32)
    p[1] = (0, 0, 1.0)
    p[2] = (0, 0, -1.0)
    return p.reshape(1, 72)


def calc_local_trans(body_model, scale=1, render_type="animate", poses=None, shape_from=None, shape_to=None, n_interp=10, max_frames=100):
    """per frame: world-space SMPL vertices [V,3] fp32 and the rest(da pose)->scene transforms [V+J,4,4] fp64 (already divided by
    SMPL_SCALE) that ray_utils.warp_samples_to_canonical inverts.  Same composition as render_warp.py:127-222, with the body model
    passed in instead of being loaded from ./data."""
    nv = body_model.v_template.shape[0]
    nj = body_model.J_regressor.shape[0]
    zero_shape = np.zeros((1, 10), np.float32)
    if render_type == "animate":
        n_frame = min(max_frames, poses.shape[0])
        target_shapes = np.zeros((n_frame, 1, 10))
    elif render_type == "interp_shape":
        target_shapes = np.linspace(shape_from, shape_to, n_interp)
        n_frame = min(max_frames, target_shapes.shape[0])
        poses = np.zeros((n_frame, 72))
    else:
        raise NotImplementedError
    da = da_pose()
    # frame-independent pieces (the reference recomputes them every frame)
    v0, T_t2rest, _ = body_model.verts_transformations(da, zero_shape, return_tensor=False, concat_joints=True)
    inv_t2rest = np.linalg.inv(T_t2rest)
    rest_v, rest_j = body_model.forward(da, zero_shape, return_joints=True, return_tensor=False)
    rest_h = np.concatenate([rest_v, rest_j], axis=0)
    rest_h = np.concatenate([rest_h, np.ones_like(rest_h[:, :1])], axis=-1)
    T_scale = np.eye(4) / SMPL_SCALE
    S = np.eye(4)
    S[:3, :3] *= scale
    world_verts, Ts = [], []
    for i in range(n_frame):
        _, T_t2pose, _ = body_model.verts_transformations(poses[i][None], zero_shape, return_tensor=False, concat_joints=True)
        vt, _, _ = body_model.verts_transformations(da, target_shapes[i], return_tensor=False, concat_joints=True)
        delta_v = (v0 - vt).squeeze()
        T_shape = np.tile(np.eye(4)[None], (nv + nj, 1, 1))
        T_shape[:, :3, 3] += delta_v
        T_rest2pose = T_t2pose @ np.linalg.inv(T_shape) @ inv_t2rest
        Ts.append(T_rest2pose @ T_scale)
        wv = np.einsum('BNi, Bi->BN', S @ T_rest2pose, rest_h)[:, :3].astype(np.float32)
        world_verts.append(wv[:nv])
    return world_verts, Ts, n_frame
