"""Is the outlier allowance of tests/parity.py justified?  `alpha >= 1/255`, `power <= 0` and `T (1 - alpha) >= 1e-4` are hard
gates on exp() outputs (forward.cu:353-361, backward.cu:562-564); any float32 evaluation of the reference algorithm flips some
(pixel, Gaussian) pairs relative to the exact result.  Measured here on BASELINE.json's C2 and C3 at full size, against the float64
build of the oracle as the exact result:

  * A = HIP vs float64 oracle,
  * B = float32 oracle (the reference algorithm in the reference's precision) vs float64 oracle.

The HIP path must be no further from the exact result than the float32 reference is: elements outside 1e-4 and the relative L2
error of every image and gradient tensor satisfy A <= 1.25 B (+ a handful of elements); and the distance HIP vs float32 oracle,
which the other tests assert, is an order of magnitude below B on the gradients.  Measured (tools/parity_stats.py,
profiles/r02/parity_stats_*.json): C3 colour 2.35e-5 vs 2.33e-5 of the elements outside 1e-4, every gradient's relative L2
identical to three digits (1.4e-4 ... 2.7e-4, dominated by the float32 `1 - T` round trip of the reference, backward.cu:449),
HIP vs float32 oracle 1.1e-5 ... 2.3e-5.  Building the blend kernels with libm's expf instead of v_exp_f32
(`make -C ad-gs_amd/csrc precise`, ADGS_LIB=...) changes none of these figures beyond the third digit and costs 4 % frames/s
(DESIGN.md section 9), so the fast exponential stays.
"""
import numpy as np
import pytest

from adgs import synthetic
from tests import parity
from tests.test_gpu_raster import run_hip, run_oracle

pytestmark = pytest.mark.gpu

PAIRS = [("means3D", "dL_dmeans3D"), ("means2D", "dL_dmeans2D"), ("opacities", "dL_dopacity"), ("shs", "dL_dsh"), ("scales", "dL_dscales"),
         ("rotations", "dL_drotations"), ("flow", "dL_dflow_points"), ("sem", "dL_dsemantic")]


@pytest.mark.parametrize("config", ["C2", "C3"])
def test_hip_is_as_close_to_the_exact_result_as_the_float32_reference(config):
    sc = synthetic.make_config_scene(config)
    g = synthetic.make_upstream_grads(sc, synthetic.CONFIGS[config]["seed"])
    h = run_hip(sc, grads=g)
    o32 = run_oracle(sc, grads=g, precision="f32")
    o64 = run_oracle(sc, grads=g, precision="f64")
    np.testing.assert_array_equal(h["radii"].cpu().numpy(), o32["radii"])
    # the float64 build rounds ceil(3 sqrt(lambda)) from a double: a handful of radii per million differ by one
    assert float((o32["radii"] != o64["radii"]).mean()) <= 2e-5 and int(np.abs(o32["radii"].astype(np.int64) - o64["radii"]).max()) <= 1
    tensors = [(k, h[k].detach().cpu().numpy(), o32[k], o64[k]) for k in ("color", "depth", "img_opacity", "img_flow", "img_semantic")]
    tensors += [("grad_" + hk, h["grads"][hk].cpu().numpy(), o32["grads"][ok], o64["grads"][ok]) for hk, ok in PAIRS]
    lines = []
    for name, a, b32, b64 in tensors:
        b32 = np.asarray(b32).reshape(a.shape); b64 = np.asarray(b64).reshape(a.shape)
        A, B, C = parity.error_stats(a, b64), parity.error_stats(b32, b64), parity.error_stats(a, b32)
        lines.append("%-16s outside 1e-4: HIP~f64 %d, f32~f64 %d, HIP~f32 %d | rel L2: %.3g, %.3g, %.3g" % (
            name, A["n_bad"], B["n_bad"], C["n_bad"], A["rel_l2"], B["rel_l2"], C["rel_l2"]))
        assert A["n_bad"] <= 1.25 * B["n_bad"] + 4, lines[-1]
        assert A["rel_l2"] <= 1.25 * B["rel_l2"] + 1e-6, lines[-1]
        assert A["max_bad_err"] <= max(1.5 * B["max_bad_err"], 2e-2 * A["scale"]), lines[-1]
        if name.startswith("grad_"):
            # the asserted distance (HIP vs float32 oracle) is well inside the float32 reference's own distance from the exact result
            assert C["rel_l2"] <= 0.25 * B["rel_l2"] + 1e-6, lines[-1]
            assert C["n_bad"] <= 0.25 * B["n_bad"] + 4, lines[-1]
    print("\n".join(lines))
