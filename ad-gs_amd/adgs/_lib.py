"""ctypes loader for libadgs_hip.so (the C ABI declared in include/adgs_rasterizer.h).

There is NO CPU fallback: if the library is missing or cannot be loaded this
module raises, and every operator built on it raises with it.
"""
import contextlib
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.dirname(_HERE)
# ADGS_LIB: another build of the same library (e.g. lib/libadgs_hip_precise.so, `make -C ad-gs_amd/csrc precise`: expf instead of
# v_exp_f32 in the blend kernels -- used by tests/test_gpu_gate_flips.py and tools/parity_stats.py)
LIB_PATH = os.environ.get("ADGS_LIB") or os.path.join(PKG_ROOT, "lib", "libadgs_hip.so")

ALLOC_FN = ctypes.CFUNCTYPE(ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t)

_lib = None

c_p = ctypes.c_void_p
c_i = ctypes.c_int
c_f = ctypes.c_float

# symbol -> (restype, argtypes); mirrors include/adgs_rasterizer.h
SIGNATURES = {
    "adgs_last_error": (ctypes.c_char_p, []),
    "adgs_device_check": (c_i, []),
    "adgs_raster_forward": (c_i, [ALLOC_FN, c_p, ALLOC_FN, c_p, ALLOC_FN, c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_i,
                                  c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_p, c_p, c_p, c_p, c_p, c_f, c_f, c_i,
                                  c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_i, c_p]),
    "adgs_raster_render": (c_i, [ALLOC_FN, c_p, ALLOC_FN, c_p, ALLOC_FN, c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_i,
                                 c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_p, c_p, c_p, c_p, c_p, c_f, c_f, c_i,
                                 c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_i, c_p]),
    "adgs_raster_render_rawsh": (c_i, [ALLOC_FN, c_p, ALLOC_FN, c_p, ALLOC_FN, c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_i,
                                       c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_p, c_p, c_p, c_p, c_f, c_f,
                                       c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_i, c_p]),
    "adgs_raster_backward": (c_i, [c_i, c_i, c_i, c_i, c_i, c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_p, c_p,
                                   c_p, c_p, c_p, c_f, c_f, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p,
                                   c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_p]),
    "adgs_raster_forward_rawsh": (c_i, [ALLOC_FN, c_p, ALLOC_FN, c_p, ALLOC_FN, c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_i,
                                        c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_p, c_p, c_p, c_p, c_f, c_f,
                                        c_p, c_p, c_p, c_p, c_p, c_i, c_p, c_i, c_p]),
    "adgs_raster_backward_rawsh": (c_i, [c_i, c_i, c_i, c_i, c_i, c_p, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_f, c_p,
                                         c_p, c_p, c_p, c_f, c_f, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p,
                                         c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_p]),
    "adgs_mark_visible": (c_i, [c_i, c_p, c_p, c_p, c_p, c_p]),
    "adgs_knn_workspace_bytes": (ctypes.c_size_t, [c_i]),
    "adgs_knn_dist2": (c_i, [c_i, c_p, c_p, c_p, c_p]),
    "adgs_get_frame_stats": (None, [c_p]),
    "adgs_get_frame_status": (c_i, [c_p]),
    "adgs_raster_needs_zero_init": (c_i, [c_i]),
    "adgs_raster_backward_needs_zero_init": (c_i, [c_p, c_p, c_i, c_i, c_i]),
    "adgs_profile_enable": (None, [c_i]),
    "adgs_profile_reserve": (c_i, [c_i]),
    "adgs_profile_num_stages": (c_i, []),
    "adgs_profile_stage_name": (ctypes.c_char_p, [c_i]),
    "adgs_profile_collect": (c_i, [c_p, c_p]),
    # include/adgs_deform.h
    "adgs_func_eval_forward": (c_i, [c_i, c_i, c_p, c_p, c_p, c_p]),
    "adgs_func_eval_backward": (c_i, [c_i, c_i, c_p, c_p, c_p, c_p, c_p]),
    "adgs_deform_forward": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "adgs_deform_backward": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    # include/adgs_envmap.h
    "adgs_envmap_forward": (c_i, [c_i, c_i, c_i, c_p, c_i, c_i, c_f, c_p, c_p, c_p]),
    "adgs_envmap_backward": (c_i, [c_i, c_i, c_i, c_i, c_i, c_f, c_p, c_p, c_p, c_p, c_p]),
    "adgs_envmap_backward_marked": (c_i, [c_i, c_i, c_i, c_i, c_i, c_f, c_p, c_p, c_p, c_p, c_p, c_i, c_p]),
    # include/adgs_loss.h
    "adgs_l1_ssim_forward": (c_i, [c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "adgs_l1_ssim_means": (c_i, [c_p, ctypes.c_longlong, c_p, c_p]),
    "adgs_l1_ssim_backward": (c_i, [c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "adgs_depth_loss_forward": (c_i, [c_i, c_p, c_p, c_p, c_p, c_p, c_p]),
    "adgs_depth_loss_backward": (c_i, [c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "adgs_flow_loss_forward": (c_i, [c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_p, c_p, c_p]),
    "adgs_flow_loss_backward": (c_i, [c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_p, c_p, c_p, c_p, c_p]),
    "adgs_flow_loss_forward_devcam": (c_i, [c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_p, c_p, c_p]),
    "adgs_flow_loss_backward_devcam": (c_i, [c_i, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_f, c_p, c_p, c_p, c_p, c_p]),
    "adgs_bce_clip_forward": (c_i, [c_i, c_p, c_p, c_f, c_f, c_i, c_i, c_p, c_p, c_p]),
    "adgs_bce_clip_backward": (c_i, [c_i, c_p, c_p, c_f, c_f, c_i, c_i, c_p, c_p, c_p]),
    "adgs_group_var_forward": (c_i, [c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p]),
    "adgs_group_var_backward": (c_i, [c_i, c_i, c_i, c_i, c_i, c_p, c_p, c_p, c_p, c_p]),
    "adgs_sigma_loss_forward": (c_i, [c_i, c_p, c_f, c_p, c_p, c_p]),
    "adgs_sigma_loss_backward": (c_i, [c_i, c_p, c_f, c_p, c_p, c_p]),
    # include/adgs_optim.h
    "adgs_densification_stats": (c_i, [c_i, c_p, c_p, c_p, c_p, c_p, c_p]),
    "adgs_adam_step": (c_i, [c_p, c_i, ctypes.c_float, ctypes.c_float, ctypes.c_float, c_i, c_p]),
    "adgs_deform_forward_flow": (c_i, [c_p] * 10),
    "adgs_deform_backward_flow": (c_i, [c_p] * 15),
    # include/adgs_exchange.h
    "adgs_sh_grad_expand": (c_i, [c_i, c_p, c_p, c_i, c_i, c_i, c_i, c_p, c_i, c_i, c_p, c_p]),
    "adgs_lin_grad_expand": (c_i, [c_i, c_p, c_p, c_i, c_i, c_f, c_p, c_p]),
    # include/adgs_densify.h
    "adgs_densify_workspace_bytes": (ctypes.c_size_t, [c_i]),
    "adgs_densify_select": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p]),
    "adgs_densify_plan": (c_i, [c_p, c_p, c_i, c_p, c_i, c_p, c_p, c_p, c_p, c_p]),
    "adgs_densify_gather_rows": (c_i, [c_p, c_p, c_i, c_i, c_p, c_p, c_i, c_p]),
    "adgs_densify_split_rows": (c_i, [c_p, c_p, c_p, c_p, c_i, c_p, c_p, c_p, c_p, c_p]),
    "adgs_reset_opacity": (c_i, [c_i, c_p, c_p]),
    # include/adgs_knn_points.h
    "adgs_knn_points_workspace_bytes": (ctypes.c_size_t, [c_i, c_i, c_i]),
    "adgs_knn_points": (c_i, [c_i, c_p, c_i, c_p, c_i, c_i, c_p, c_p, c_p, c_p]),
    # include/adgs_testing.h
    "adgs_test_v2_published_entries": (ctypes.c_longlong, [c_p, c_i, c_i, c_p]),
    "adgs_test_v2_scanned_candidates": (ctypes.c_longlong, [c_p, c_i, c_i, c_p]),
    "adgs_test_v2_tile_counters": (ctypes.c_longlong, [c_p, c_i, c_i, c_p, c_p, ctypes.c_longlong, c_p]),
    "adgs_test_v2_blend_batches": (ctypes.c_longlong, [c_p, c_i, c_i, c_p]),
    "adgs_test_v2_cell_ranges": (ctypes.c_longlong, [c_p, c_i, c_i, c_p, ctypes.c_longlong, c_p]),
    "adgs_test_abi_sizeof": (ctypes.c_size_t, [c_i]),
    "adgs_test_set_capacity_hints": (None, [ctypes.c_longlong, ctypes.c_longlong]),
    "adgs_test_env_reads": (ctypes.c_ulonglong, []),
    "adgs_test_scramble_slab_bounds": (c_i, [ctypes.c_uint]),
    "adgs_test_scan_temp_bytes": (ctypes.c_size_t, [ctypes.c_size_t]),
    "adgs_test_exclusive_scan_u32": (c_i, [c_p, c_p, ctypes.c_size_t, c_p, c_p]),
    "adgs_test_sort_temp_bytes": (ctypes.c_size_t, [ctypes.c_size_t]),
    "adgs_test_sort_pairs_u64": (c_i, [c_p, c_p, c_p, c_p, ctypes.c_size_t, c_i, c_p, c_p]),
    "adgs_test_sort_pairs_u32": (c_i, [c_p, c_p, c_p, c_p, ctypes.c_size_t, c_i, c_p, c_p]),
}


class FrameStats(ctypes.Structure):
    _fields_ = [("num_rendered", ctypes.c_int64), ("tiles", ctypes.c_int32), ("sort_bits", ctypes.c_int32),
                ("sort_passes", ctypes.c_int32), ("reserved", ctypes.c_int32), ("fine_pairs", ctypes.c_int64)]


class FrameStatus(ctypes.Structure):
    """adgs_frame_status (include/adgs_rasterizer.h)."""
    _fields_ = [("pairs", ctypes.c_int64), ("fine_pairs", ctypes.c_int64), ("capacity_pairs", ctypes.c_int64),
                ("capacity_fine_pairs", ctypes.c_int64), ("overflow_count", ctypes.c_int64), ("eager_reruns", ctypes.c_int64),
                ("overflow", ctypes.c_int32), ("order_hint", ctypes.c_int32), ("unrepaired_overflow_count", ctypes.c_int64),
                ("order_hint_lookups", ctypes.c_int64), ("order_hint_hits", ctypes.c_int64), ("fullest_slab_units", ctypes.c_int64)]


def lib():
    """Load (once) and return the ctypes handle; raises if the HIP library is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libadgs_hip.so not found at %s: build it with `make -C ad-gs_amd/csrc` "
                "(or __graft_entry__.build()). There is no CPU fallback." % LIB_PATH)
        # torch first: this library and torch must share ONE HIP runtime, and it has to be the one torch ships -- with the
        # system's libamdhip64 loaded first (this library's own dependency), torch.cuda.is_available() turns False and the two
        # sides stop seeing each other's devices and streams (observed on the MI355X boxes: "no HIP device" from adgs_device_check
        # in a process that had built and loaded the library before importing torch)
        import torch  # noqa: F401
        handle = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)      # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def last_error():
    msg = lib().adgs_last_error()
    return msg.decode("utf-8", "replace") if msg else ""


_NULL_CTX = contextlib.nullcontext()


def on_device(device):
    """`with torch.cuda.device(device)` only when `device` is not already the current one (the context manager costs ~10 us per use and
    every wrapper of the library needs the right device current while it launches)."""
    import torch
    return _NULL_CTX if torch._C._cuda_getDevice() == device.index else torch.cuda.device(device)


def stream_ptr(device):
    """The current HIP stream of `device` as the `void* stream` argument of the C ABI.  torch.cuda.current_stream() builds a Stream object
    per call (~5 us, and every wrapper of an iteration asks: 19 times per training iteration); the raw handle is one C call."""
    import torch
    return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(device.index if device.index is not None else torch._C._cuda_getDevice()))


def check(code, what):
    if code < 0:
        raise RuntimeError("%s failed: %s" % (what, last_error()))
    return code


def frame_stats():
    st = FrameStats()
    lib().adgs_get_frame_stats(ctypes.byref(st))
    return dict(num_rendered=int(st.num_rendered), tiles=int(st.tiles), sort_bits=int(st.sort_bits), sort_passes=int(st.sort_passes),
                fine_pairs=int(st.fine_pairs), bucket_binning=int(st.reserved))


def frame_status():
    """Capacity status of this thread's most recent default-pipeline forward on the current device (adgs_get_frame_status):
    call after the stream has been synchronised.  `overflow` / `overflow_count` matter for frames replayed from a HIP graph
    (adgs.graph.GraphedStep); eager forwards repair themselves before they return (`eager_reruns`)."""
    st = FrameStatus()
    check(lib().adgs_get_frame_status(ctypes.byref(st)), "adgs_get_frame_status")
    return {k: int(getattr(st, k)) for k, _ in FrameStatus._fields_ if k != "reserved"}


class StageProfiler:
    """Per-stage HIP-event timing of the native pipeline (bench.py)."""

    def __init__(self):
        self.lib = lib()
        self.n = self.lib.adgs_profile_num_stages()
        self.names = [self.lib.adgs_profile_stage_name(i).decode() for i in range(self.n)]
        self.total = (ctypes.c_double * self.n)()
        self.count = (ctypes.c_int64 * self.n)()

    def reserve(self, n_events):
        """Pre-create HIP events (two per timed launch group and step)."""
        return self.lib.adgs_profile_reserve(int(n_events))

    def enable(self, on=True, stages=None):
        """stages: iterable of stage names to time (default: all).  Every timed stage costs two event records per launch
        group, i.e. a ~10 us bubble in the queue: time only what you report inside a throughput measurement."""
        mask = 0
        if on:
            mask = -1 if stages is None else sum(1 << self.names.index(s) for s in stages)
        self.lib.adgs_profile_enable(mask)

    def collect(self):
        """Call after torch.cuda.synchronize(); returns {stage: (avg_ms, launches)}."""
        self.lib.adgs_profile_collect(self.total, self.count)
        return {self.names[i]: ((self.total[i] / self.count[i]) if self.count[i] else 0.0, int(self.count[i])) for i in range(self.n)}
