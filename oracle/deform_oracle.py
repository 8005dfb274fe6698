"""NumPy restatement of the reference's per-frame deformation (CPU ORACLE -- TEST
INFRASTRUCTURE ONLY; see oracle/raster_oracle.cpp header for the import rules).

Follows utils/func_utils.py:33-173 (`get_deboor_cox_mat`, basis functions,
`get_func_result`) and scene/gaussian_model.py:89-231 (`get_deformed_*`,
`get_time_masked_opacity`, activations :36-44) of the reference.

PARITY STATUS
  * B-spline / polynomial / Fourier families and get_deformed_pkg: PINNED against the
    reference Python itself, imported in the authoring container
    (tests/golden/make_deform_golden.py -> tests/golden/deform_*.npz).
  * Quaternion B-spline: the reference calls `roma==1.5.1` (environment.yaml:261;
    unitquat_to_rotvec / rotvec_to_unitquat / quat_product / quat_conjugation, XYZW),
    which is not installable here -> "parity unpinned" at that boundary.  The four
    functions are restated below from the library's published algorithm (shortest-arc
    log with the small-angle series 2 + a^2/12 + 7a^4/2880, exp with
    1/2 - n^2/48 + n^4/3840, threshold 1e-3) and cross-checked against
    scipy.spatial.transform.Rotation in tests/test_oracle_deform.py.
"""
import numpy as np

F32 = np.float32


def get_deboor_cox_mat(order):
    """utils/func_utils.py:33-50 (float32 recursion, returned as float32)."""
    if order == 0:
        return np.array([[1.0]], dtype=F32)
    prior = get_deboor_cox_mat(order - 1)
    left = np.concatenate([prior, np.zeros((1, prior.shape[1]), dtype=F32)], axis=0)
    right = np.concatenate([np.zeros((1, prior.shape[1]), dtype=F32), prior], axis=0)
    tl = np.zeros((order, order + 1), dtype=F32)
    idx = np.arange(order, dtype=np.int32)
    tl[idx, idx] = idx + 1
    tl[idx, idx + 1] = order - idx - 1
    tr = np.zeros((order, order + 1), dtype=F32)
    tr[idx, idx] = -1
    tr[idx, idx + 1] = 1
    return ((left @ tl + right @ tr) / order).astype(F32)


def bspline_basis(u, order, dtype=F32):
    """utils/func_utils.py:65-77: (u ** [0..k]) @ M_k ; 0**0 == 1."""
    freq = np.arange(0, order + 1, dtype=dtype)
    pw = np.power(dtype(u), freq).astype(dtype)
    return (pw @ get_deboor_cox_mat(order).astype(dtype)).astype(dtype)


def fft_basis(v, order, dtype=F32):
    """utils/func_utils.py:52-57: [sin(f*pi*v) f=1..F | cos(f*pi*v)]."""
    freq = (np.linspace(1.0, order, order).astype(dtype) * dtype(np.pi)).astype(dtype)
    x = (dtype(v) * freq).astype(dtype)
    return np.concatenate([np.sin(x), np.cos(x)]).astype(dtype)


def poly_basis(v, order, dtype=F32):
    """utils/func_utils.py:59-63: v ** [1..n]."""
    freq = np.linspace(1.0, order, order).astype(dtype)
    return np.power(dtype(v), freq).astype(dtype)


def get_param_num(args):
    return args[0] + args[2] + 2 * args[3] + args[4]


def segment(v, nctrl, k):
    """Segment index / local parameter of the uniform B-spline (func_utils.py:128-131, 157-161)."""
    interval = nctrl - k
    start = min(int(v * interval), interval - 1)
    u = v * interval - start
    return start, u


# ---------------- roma (XYZW) restatement -----------------
def quat_conjugation(q):
    out = q.copy()
    out[..., :3] *= -1
    return out


def quat_product(p, q):
    vec = p[..., 3:4] * q[..., :3] + q[..., 3:4] * p[..., :3] + np.cross(p[..., :3], q[..., :3])
    last = p[..., 3] * q[..., 3] - np.sum(p[..., :3] * q[..., :3], axis=-1)
    return np.concatenate([vec, last[..., None]], axis=-1)


def unitquat_to_rotvec(q):
    q = q.copy()
    neg = q[..., 3] < 0
    q[neg] *= -1                                   # shortest arc
    angle = 2 * np.arctan2(np.linalg.norm(q[..., :3], axis=-1), q[..., 3])
    small = angle <= 1e-3
    scale = np.empty_like(angle)
    a = angle[small]
    scale[small] = 2 + a ** 2 / 12 + 7 * a ** 4 / 2880
    a = angle[~small]
    scale[~small] = a / np.sin(a / 2)
    return scale[..., None] * q[..., :3]


def rotvec_to_unitquat(rv):
    n = np.linalg.norm(rv, axis=-1)
    small = n <= 1e-3
    scale = np.empty_like(n)
    s = n[small]
    scale[small] = 0.5 - s ** 2 / 48 + s ** 4 / 3840
    s = n[~small]
    scale[~small] = np.sin(s / 2) / s
    return np.concatenate([scale[..., None] * rv, np.cos(n / 2)[..., None]], axis=-1)


def get_func_result(v, param, order_args, dtype=F32):
    """utils/func_utils.py:121-173.  param [..., D, n_params]; returns [..., D] or the python
    float 0.0 when every order is zero (func_utils.py:123)."""
    param = np.asarray(param, dtype=dtype)
    result = 0.0
    offset = 0
    if order_args[0] != 0:
        start, u = segment(v, order_args[0], order_args[1])
        ctrl = param[..., start + offset: start + order_args[1] + offset + 1]
        func = bspline_basis(u, order_args[1], dtype)
        result = result + np.sum(ctrl * func, axis=-1, dtype=dtype)
        offset += order_args[0]
    if order_args[2] != 0:
        p = param[..., offset: offset + order_args[2]]
        result = result + np.sum(p * poly_basis(v, order_args[2], dtype), axis=-1, dtype=dtype)
        offset += order_args[2]
    if order_args[3] != 0:
        p = param[..., offset: offset + order_args[3] * 2]
        result = result + np.sum(p * fft_basis(v, order_args[3], dtype), axis=-1, dtype=dtype)
        offset += order_args[3] * 2
    if order_args[4] != 0:
        start, u = segment(v, order_args[4], order_args[5])
        k = order_args[5]
        ctrl = param[..., start + offset: start + k + offset + 1] + np.array([1.0, 0.0, 0.0, 0.0], dtype=dtype).reshape(-1, 1)
        ctrl = np.transpose(ctrl, (0, 2, 1))                        # N, k+1, 4 (wxyz)
        ctrl = ctrl / np.maximum(np.linalg.norm(ctrl, axis=-1, keepdims=True), dtype(1e-12))
        ctrl = ctrl[..., [1, 2, 3, 0]]                              # xyzw
        func = bspline_basis(u, k, dtype)
        func_cum = np.flip(np.cumsum(np.flip(func), dtype=dtype))[1:]
        conj = quat_conjugation(ctrl[:, :-1, :])
        vec = unitquat_to_rotvec(quat_product(conj, ctrl[:, 1:, :]))
        quat = rotvec_to_unitquat(vec * func_cum[None, :, None])
        vector = ctrl[:, 0]
        for i in range(quat.shape[1]):
            vector = quat_product(vector, quat[:, i])
        result = result + vector[..., [3, 0, 1, 2]].astype(dtype)
        offset += order_args[4]
    return result


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def normalize(x, eps=1e-12):
    return x / np.maximum(np.linalg.norm(x, axis=-1, keepdims=True), eps)


def get_deformed_pkg(m, t, dtype=F32):
    """scene/gaussian_model.py:173-231 on a dict `m` of numpy arrays named like the
    reference's attributes (scene_xyz, obj_xyz, ..., order_args, use_time_mask)."""
    oa = m["order_args"]
    f = lambda k: np.asarray(m[k], dtype=dtype)
    obj_xyz = f("obj_xyz") + get_func_result(t, f("xyz_deform_param"), oa["xyz"], dtype)
    xyz = np.concatenate([f("scene_xyz"), obj_xyz], 0)
    xyz = xyz + get_func_result(t, f("background_deform_param"), oa["background"], dtype)
    obj_rot = get_func_result(t, f("rotation_deform_param"), oa["rotation"], dtype)
    if oa["rotation"][4] == 0:
        obj_rot = f("obj_rotation") + obj_rot
    rotation = normalize(np.concatenate([f("scene_rotation"), obj_rot], 0).astype(dtype))
    shs_param = np.concatenate([f("shs_deform_param_scene"), f("shs_deform_param_obj")], 0)
    shs_dc = np.concatenate([f("scene_shs_dc"), f("obj_shs_dc")], 0)[:, 0] + get_func_result(t, shs_param, oa["shs"], dtype)
    shs_rest = np.concatenate([f("scene_shs_rest"), f("obj_shs_rest")], 0)
    shs = np.concatenate([shs_dc[:, None], shs_rest], 1)
    if m.get("use_time_mask"):
        dt = dtype(t) - f("gs_time")
        sig = np.exp(f("gs_time_sigma"))
        sig = np.where(dt < 0.0, sig[:, :1], sig[:, 1:])
        mask = np.exp(-0.5 * (dt / sig) ** 2)
        opacity = np.concatenate([sigmoid(f("scene_opacity")), sigmoid(f("obj_opacity")) * mask], 0)
    else:
        opacity = sigmoid(np.concatenate([f("scene_opacity"), f("obj_opacity")], 0))
    scales = np.exp(np.concatenate([f("scene_scaling"), f("obj_scaling")], 0))
    return dict(xyz=xyz.astype(dtype), rotation=rotation.astype(dtype), shs=shs.astype(dtype), opacity=opacity.astype(dtype),
                scales=scales.astype(dtype))
