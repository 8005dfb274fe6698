"""HIP-graph replay of a whole frame (deformation -> preprocess -> binning -> blend -> backward chain).

The reference drains the stream in the middle of every forward to size its binning buffer
(submodules/depth-diff-gaussian-rasterization/cuda_rasterizer/rasterizer_impl.cu:288), so its frame cannot be captured at all.
This library enqueues the whole forward against a capacity and validates it afterwards (include/adgs_rasterizer.h:
adgs_get_frame_status), which makes a frame a fixed launch sequence: ~20 kernel launches, a dozen allocator calls and two
autograd-function dispatches become one `hipGraphLaunch`.  That is what a launch-bound frame (C1: 10 k Gaussians at 400x300 is
tens of microseconds of GPU work under 0.3 ms of Python) needs, and what removes the launch gaps of a large one.

    step = GraphedStep(fn)        # fn(): forward + backward on STATIC inputs (parameters, camera, upstream gradients in place)
    step()                        # replay
    step.validate()               # after a synchronisation: did every replay since the last check fit its capacity?
                                  # A replay that did not fit is a defined EMPTY render (background colour, no gradients); validate
                                  # BEFORE the optimizer consumes its gradients, or use step.checked() which does both.

Rules (torch.cuda.graphs): nothing may keep the autograd graph of an EARLIER eager call of fn alive (a retained loss or output
with a grad_fn pins the leaves' AccumulateGrad nodes to the stream of that call, and the capture then crosses streams: detach
what you keep); fn must not synchronise, must read its inputs from tensors that keep their addresses, and leaves its
results (parameter .grad, returned tensors) in the same tensors at every replay.  Basis values of the deformation functions and
camera constants are launch arguments, i.e. baked into the captured graph: one GraphedStep per (camera, time stamp); a change
of the number of Gaussians (densification) needs a new capture.  `GraphCache` keeps one step per key on a shared memory pool.
"""
import torch

from . import _lib


class GraphedStep:
    def __init__(self, fn, warmup=2, pool=None):
        """Runs fn `warmup` times eagerly on a side stream (this also lets the eager path measure the frame and raise the capacity
        hints the capture is enqueued against), then captures it."""
        self.fn = fn
        self.recaptures = 0
        self._pool = pool
        self._seen_overflows = None
        self._capture(warmup)

    def _capture(self, warmup):
        torch.cuda.synchronize()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(max(warmup, 1)):
                self.fn()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, pool=self._pool):
            self.result = self.fn()
        torch.cuda.synchronize()
        self._seen_overflows = _lib.frame_status()["unrepaired_overflow_count"]

    def pool(self):
        return self.graph.pool()

    def __call__(self):
        self.graph.replay()
        return self.result

    def checked(self):
        """Replay, synchronise, validate: the results, or None if this replay did not fit (the step has been re-captured; call again)."""
        out = self()
        torch.cuda.synchronize()
        return out if self.validate() else None

    def validate(self, repair=True):
        """Call with the stream synchronised.  True: every replay since the last call fitted the capacity it was captured with.
        False: at least one did not -- its results are invalid; with `repair` the step has been re-captured (an eager frame first,
        which raises the capacity hints) and the caller must redo the affected steps."""
        st = _lib.frame_status()
        ok = st["unrepaired_overflow_count"] == self._seen_overflows      # eager frames (warm-ups of other captures, evaluation renders) repair themselves and do not count
        self._seen_overflows = st["unrepaired_overflow_count"]
        if not ok and repair:
            self.recaptures += 1
            self._capture(1)
        return ok


class GraphCache:
    """One GraphedStep per key (e.g. camera id, time stamp), captured on first use, all on one memory pool: only one of them
    runs at a time, so their temporaries may share memory; what a step RETURNS stays referenced and therefore private."""

    def __init__(self, make_fn, warmup=2):
        self.make_fn, self.warmup = make_fn, warmup
        self.steps, self._pool, self._used = {}, None, set()
        self._seen_overflows = None
        self.recaptures = 0

    def __call__(self, key):
        step = self.steps.get(key)
        if step is None:
            step = self.steps[key] = GraphedStep(self.make_fn(key), warmup=self.warmup, pool=self._pool)
            if self._pool is None:
                self._pool = step.pool()
            if self._seen_overflows is None:
                self._seen_overflows = _lib.frame_status()["unrepaired_overflow_count"]
        self._used.add(key)
        return step()

    def validate(self, repair=True):
        """Call with the stream synchronised.  The overflow counter is shared by every frame of this thread and device, so a change
        condemns every key replayed since the last call: True = all of them fitted; False = they are re-captured (with `repair`)
        and the caller must redo those steps."""
        now = _lib.frame_status()["unrepaired_overflow_count"]
        ok = self._seen_overflows is None or now == self._seen_overflows
        if not ok and repair:
            for key in self._used:
                self.steps[key]._capture(1)
                self.recaptures += 1
            now = _lib.frame_status()["unrepaired_overflow_count"]
        self._seen_overflows = now
        self._used = set()
        return ok

    def clear(self):
        self.steps.clear()
        self._pool, self._used = None, set()
