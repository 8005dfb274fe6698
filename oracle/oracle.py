"""ctypes/numpy binding of the CPU oracle.

TEST INFRASTRUCTURE ONLY: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module (see oracle/raster_oracle.cpp header).
The product path (ad-gs_amd/) never imports it.

The argument order of RasterOracle.forward/backward mirrors the reference's
`_C.rasterize_gaussians` / `_C.rasterize_gaussians_backward`
(submodules/depth-diff-gaussian-rasterization/rasterize_points.cu:35-140,142-254).
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libadgs_oracle.so")
_lib = None


def build(force=False):
    """Compile oracle/*.cpp with g++ (see oracle/Makefile)."""
    srcs = [os.path.join(_HERE, f) for f in ("raster_oracle.cpp", "knn_oracle.cpp", "Makefile")]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in srcs)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "libadgs_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        for suf in ("f32", "f64"):
            getattr(_lib, "adgs_oracle_create_" + suf).restype = ctypes.c_void_p
        _lib.adgs_oracle_get_higher_msb.restype = ctypes.c_uint32
    return _lib


def _f32(a, shape=None):
    if a is None:
        return None
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float32))
    if a.size == 0:
        return None
    return a


def _ptr(a):
    if a is None:
        return ctypes.c_void_p(0)
    return ctypes.c_void_p(a.ctypes.data)


def set_num_threads(n):
    lib().adgs_oracle_set_num_threads(int(n))


def num_threads():
    return int(lib().adgs_oracle_num_threads())


def get_higher_msb(n):
    return int(lib().adgs_oracle_get_higher_msb(ctypes.c_uint32(n)))


class RasterOracle:
    """Stateful forward/backward pair (state = the reference's geom/binning/img buffers)."""

    def __init__(self, precision="f32"):
        assert precision in ("f32", "f64")
        self.suf = precision
        self.real = np.float32 if precision == "f32" else np.float64
        self._h = ctypes.c_void_p(getattr(lib(), "adgs_oracle_create_" + self.suf)())
        self.P = 0

    def __del__(self):
        try:
            getattr(lib(), "adgs_oracle_destroy_" + self.suf)(self._h)
        except Exception:
            pass

    def forward(self, bg, means3D, colors_precomp, opacities, scales, rotations, scale_modifier, cov3D_precomp,
                viewmatrix, projmatrix, tanfovx, tanfovy, image_height, image_width, sh, flow_points, semantic,
                degree, campos, prefiltered=False, inv_depth=False):
        means3D = _f32(means3D)
        if means3D is None:
            means3D = np.zeros((0, 3), np.float32)
        if means3D.ndim != 2 or means3D.shape[1] != 3:
            raise ValueError("means3D must have dimensions (num_points, 3)")
        P = means3D.shape[0]
        H, W = int(image_height), int(image_width)
        sh = _f32(sh)
        colors_precomp = _f32(colors_precomp)
        flow_points = _f32(flow_points)
        semantic = _f32(semantic)
        scales = _f32(scales)
        rotations = _f32(rotations)
        cov3D_precomp = _f32(cov3D_precomp)
        opacities = _f32(opacities)
        M = sh.shape[1] if sh is not None else 0
        D_S = semantic.shape[1] if semantic is not None else 0
        assert D_S <= 32
        if flow_points is not None:
            assert flow_points.shape[1] == 3
        bg = _f32(bg)
        view = _f32(viewmatrix)
        proj = _f32(projmatrix)
        campos = _f32(campos)
        r = self.real
        out = dict(
            color=np.zeros((3, H, W), r), depth=np.zeros((1, H, W), r), img_opacity=np.zeros((1, H, W), r),
            img_flow=np.zeros((3, H, W), r), img_semantic=np.zeros((D_S, H, W), r), radii=np.zeros((P,), np.int32))
        self.P, self.M, self.D_S, self.H, self.W = P, M, D_S, H, W
        self._keep = (bg, means3D, sh, colors_precomp, flow_points, semantic, opacities, scales, rotations, cov3D_precomp, view, proj, campos)
        R = 0
        if P != 0:
            R = getattr(lib(), "adgs_oracle_forward_" + self.suf)(
                self._h, P, int(degree), M, D_S, _ptr(bg), W, H, _ptr(means3D), _ptr(sh), _ptr(colors_precomp), _ptr(flow_points),
                _ptr(semantic), _ptr(opacities), _ptr(scales), ctypes.c_float(scale_modifier), _ptr(rotations), _ptr(cov3D_precomp),
                _ptr(view), _ptr(proj), _ptr(campos), ctypes.c_float(tanfovx), ctypes.c_float(tanfovy),
                _ptr(out["color"]), _ptr(out["depth"]), _ptr(out["img_opacity"]), _ptr(out["img_flow"]), _ptr(out["img_semantic"]),
                int(bool(inv_depth)), _ptr(out["radii"]))
        out["num_rendered"] = int(R)
        self.R = int(R)
        return out

    def gate_margins(self):
        """After forward(): how close every pixel's walk came to one of the three hard gates of forward.cu:345-361, in units of the
        rounding error of a float32 evaluation (raster_oracle.cpp: gate_margins), and per Gaussian the minimum over the pixels it
        takes part in.  tests/parity.py uses them to tell a gate flip from a wrong result."""
        pix = np.full((self.H, self.W), 1e30, self.real)
        gauss = np.full((self.P,), 1e30, self.real)
        if self.P != 0:
            getattr(lib(), "adgs_oracle_gate_margins_" + self.suf)(self._h, _ptr(pix), _ptr(gauss))
        cond = np.zeros((self.H, self.W), self.real)
        if self.P != 0:
            getattr(lib(), "adgs_oracle_pixel_conditioning_" + self.suf)(self._h, _ptr(cond))
        return dict(pixel=pix, gauss=gauss, cond=cond)

    def state(self):
        """Internal buffers (geometry / binning / image state) for white-box tests."""
        P, R, r = self.P, self.R, self.real
        T = ((self.W + 15) // 16) * ((self.H + 15) // 16)
        st = dict(means2D=np.zeros((P, 2), r), depths=np.zeros((P,), r), cov3D=np.zeros((P, 6), r), rgb=np.zeros((P, 3), r),
                  conic_opacity=np.zeros((P, 4), r), clamped=np.zeros((P, 3), np.uint8), tiles_touched=np.zeros((P,), np.uint32),
                  point_list=np.zeros((R,), np.uint32), keys=np.zeros((R,), np.uint64), ranges=np.zeros((T, 2), np.uint32),
                  n_contrib=np.zeros((self.H, self.W), np.uint32))
        if P != 0:
            getattr(lib(), "adgs_oracle_get_state_" + self.suf)(
                self._h, _ptr(st["means2D"]), _ptr(st["depths"]), _ptr(st["cov3D"]), _ptr(st["rgb"]), _ptr(st["conic_opacity"]),
                _ptr(st["clamped"]), _ptr(st["tiles_touched"]), _ptr(st["point_list"]), _ptr(st["keys"]), _ptr(st["ranges"]),
                _ptr(st["n_contrib"]))
        return st

    def backward(self, grad_color, grad_depth, grad_flow, grad_semantic, grad_img_opacity):
        P, M, D_S, r = self.P, self.M, self.D_S, self.real
        g = dict(
            dL_dmeans2D=np.zeros((P, 3), r), dL_dconic=np.zeros((P, 2, 2), r), dL_dopacity=np.zeros((P, 1), r),
            dL_dcolors=np.zeros((P, 3), r), dL_ddepths=np.zeros((P, 1), r), dL_dmeans3D=np.zeros((P, 3), r),
            dL_dcov3D=np.zeros((P, 6), r), dL_dsh=np.zeros((P, M, 3), r), dL_dscales=np.zeros((P, 3), r),
            dL_drotations=np.zeros((P, 4), r), dL_dflow_points=np.zeros((P, 3), r), dL_dsemantic=np.zeros((P, D_S), r))
        if P == 0:
            return g
        gc, gd, gf, gs, go = (_f32(x) for x in (grad_color, grad_depth, grad_flow, grad_semantic, grad_img_opacity))
        getattr(lib(), "adgs_oracle_backward_" + self.suf)(
            self._h, _ptr(gc), _ptr(gd), _ptr(gf), _ptr(gs), _ptr(go),
            _ptr(g["dL_dmeans2D"]), _ptr(g["dL_dconic"]), _ptr(g["dL_dopacity"]), _ptr(g["dL_dcolors"]), _ptr(g["dL_ddepths"]),
            _ptr(g["dL_dmeans3D"]), _ptr(g["dL_dcov3D"]), _ptr(g["dL_dsh"]), _ptr(g["dL_dscales"]), _ptr(g["dL_drotations"]),
            _ptr(g["dL_dflow_points"]), _ptr(g["dL_dsemantic"]))
        return g

    def mark_visible(self, means3D, viewmatrix, projmatrix):
        means3D = _f32(means3D)
        P = 0 if means3D is None else means3D.shape[0]
        present = np.zeros((P,), np.uint8)
        if P:
            getattr(lib(), "adgs_oracle_mark_visible_" + self.suf)(self._h, P, _ptr(means3D), _ptr(_f32(viewmatrix)), _ptr(_f32(projmatrix)), _ptr(present))
        return present.astype(bool)


def knn_dist2(points, bruteforce=False):
    """Oracle for simple_knn._C.distCUDA2 (KNN/spatial.cu:15-26)."""
    pts = np.ascontiguousarray(np.asarray(points, dtype=np.float32))
    P = pts.shape[0]
    out = np.zeros((P,), np.float32)
    if P:
        fn = lib().adgs_oracle_knn_bruteforce if bruteforce else lib().adgs_oracle_knn
        fn(P, _ptr(pts), _ptr(out))
    return out
