"""Frame-at-a-time live-stream session (BASELINE.json config 5; reference loop: demo.py:238-252).

The reference's demo slides a `seqlen` window over one tracked person: window k holds the features of frames k .. k+seqlen-1
and, in its theta slots, the predictions of the first seqlen-1 of them (the newest frame's slots are zero); the prediction for
the newest frame is appended to the theta history and the window moves on by one frame.  `tepose_amd.driver.run_clips` runs
that loop when the whole clip is known up front.  A live stream is not: frames ARRIVE, one at a time, and what matters is the
time from a frame's arrival on the host to its result being readable on the host.

`StreamSession.push(feature[2048])` is that step.  The session owns the window on the device (features + theta history), a
pinned staging row for the arriving feature and pinned result rows; one push is ONE captured hipGraph replay --

    H2D copy of the feature -> shift the window by one frame, newest theta slots zero -> TePose.forward (B = 1: the
    persistent small-batch HIP kernels) -> predicted theta into the history slot -> D2H copy of the kept outputs

-- followed by one event wait.  No allocation, no Python-side tensor work and no host sync inside the step; the host pays
one graph launch per frame instead of ~10 kernel launches plus the slicing of `run_clips`.  The forward inside the graph is
the same C entry point on the same operands as the eager driver, so the results are bit-identical to `run_clips`
(tests/test_gpu_stream.py).

Failure contract (include/tepose_amd.h, "failure channel"): a replayed graph is not status-checked by `Engine._run`, so
`push` reads the handle's fault word (one host-memory read) after its wait; on a give-up the model is switched to the
step-per-launch kernels, the graph is re-captured and the frame is recomputed from the saved window -- never a NaN result.
"""

import torch

from . import _lib


class StreamSession:
    """One live clip.  `theta_init` [seqlen-1, 85] and `feature_init` [seqlen-1, 2048] are the history a stream starts from
    (demo.py:229-237 takes both from the VIBE bootstrap over the first frames).  `push(feature)` returns a dict of pinned host
    tensors for the newest frame (valid until the next push): keys `keep`, e.g. theta[85], kp_3d[J,3], verts[6890,3]."""

    def __init__(self, model, seqlen, feature_init, theta_init, J_regressor=None, keep=('theta', 'kp_3d', 'verts'), graph=True):
        self.model, self.T, self.J, self.keep = model, int(seqlen), J_regressor, tuple(keep)
        T = self.T
        dev = next(model.parameters()).device
        self.dev = dev
        if T < 2 or tuple(feature_init.shape) != (T - 1, 2048) or tuple(theta_init.shape) != (T - 1, 85):
            raise ValueError('feature_init / theta_init must be [seqlen-1, 2048] / [seqlen-1, 85] with seqlen >= 2')
        self.stream = torch.cuda.Stream(device=dev)
        eng = model._engine
        with torch.no_grad():
            eng.pack_model(model, dev)          # the workspace size depends on what is packed
        # A workspace of the session's own: the captured graph bakes its address into every kernel argument (the status words
        # that push() reads live in it too), and the engine's shared workspace moves whenever another call on the model needs
        # more bytes or the weights are re-packed.  Two sessions on one model do not share scratch either.
        self.ws = torch.empty(eng.workspace_bytes(1, T), dtype=torch.uint8, device=dev)
        self._captured_for = None
        self._rewatch()
        # the window as the model sees it (slot T-1 = the newest frame) and a scratch copy for the one-frame shift
        self.win = torch.zeros(1, T, 2133, device=dev)
        self.win[0, 1:, :2048] = feature_init.to(dev, torch.float32)      # after the first push's shift they sit in slots 0 .. T-2
        self.win[0, 1:, 2048:] = theta_init.to(dev, torch.float32)
        self.tmp = torch.empty_like(self.win)
        self.saved = torch.empty_like(self.win)                           # the window before a push (recompute after a fault)
        self.feat_host = torch.zeros(2048, dtype=torch.float32).pin_memory()
        self.out_host = {}
        self.graph = None
        self.use_graph = bool(graph)
        self.done = torch.cuda.Event()
        self.frames = 0
        # the buffers above were filled on the caller's stream; the session's stream is non-blocking (no implicit ordering)
        self.stream.wait_stream(torch.cuda.current_stream(dev))
        with torch.no_grad(), torch.cuda.stream(self.stream), eng.use_workspace(self.ws):
            # warm-up (its window is restored afterwards): packs the blob, sizes the workspace, creates the pinned result rows
            for attempt in (0, 1):
                with model._engine.lazy_status():
                    self._step()
                self.stream.synchronize()
                try:
                    model._engine.check_status()             # a give-up here has switched the model to the step kernels (warned)
                    break
                except _lib.TeposeTimeout:
                    if attempt:
                        raise
                finally:
                    self.win.copy_(self.saved)
            if self.use_graph:
                self._capture()
            else:
                self._captured_for = self._signature()
        self.stream.synchronize()

    # one step, queued on the current stream: what the graph captures
    def _step(self):
        T = self.T
        self.saved.copy_(self.win)                                   # the recompute point, should this step give up
        # shift by one frame (slots 1 .. T-1 -> 0 .. T-2, through tmp: the ranges overlap); the arriving feature goes into slot
        # T-1 straight from the pinned staging row, its theta slots are zero (evaluate.py:248-252, demo.py:239-241)
        self.tmp[0, :T - 1].copy_(self.win[0, 1:])
        self.tmp[0, T - 1, :2048].copy_(self.feat_host, non_blocking=True)
        self.tmp[0, T - 1, 2048:].zero_()
        self.win.copy_(self.tmp)
        out = self.model(self.win, J_regressor=self.J)[0]
        self.win[0, T - 1, 2048:].copy_(out['theta'][0])             # history: this frame's theta, in a window from the next push on
        for k in self.keep:
            if k not in self.out_host:
                self.out_host[k] = torch.empty(out[k][0].shape, dtype=torch.float32).pin_memory()
            self.out_host[k].copy_(out[k][0], non_blocking=True)

    def _rewatch(self):
        eng = self.model._engine
        self._watch = eng._enc_tensors(self.model.encoder) + eng._reg_tensors(self.model.regressor)
        self._watch_sig = (tuple([t._version for t in self._watch]), self._watch[0].data_ptr(), self._watch[-1].data_ptr())

    def _signature(self):
        eng = self.model._engine
        return (eng.packed_generation, eng.blob.data_ptr() if eng.blob is not None else 0, self.ws.data_ptr())

    def _refresh(self):
        """Before a step: re-pack if the model's parameters changed in place since the last pack (outside any capture -- a replay would go on
        using the old packed blob without a sign), grow the session's workspace if the packed model now needs more, and re-capture a graph
        whose blob / workspace are no longer the ones it was captured against.  Same for the eager (graph=False) session."""
        eng = self.model._engine
        # Per push, on the frame's critical path: the in-place version counters of the watched parameter / buffer objects and two addresses (~5 us on the
        # host).  The engine's full signature walk (every tensor re-fetched from its module by name, ~35 us) runs when those moved, and every 32nd push as a
        # backstop for what the cheap check cannot see (a parameter OBJECT replaced by assignment).
        self._pushes_since_full = getattr(self, '_pushes_since_full', 0) + 1
        cheap = (tuple([t._version for t in self._watch]), self._watch[0].data_ptr(), self._watch[-1].data_ptr())
        if cheap != self._watch_sig or self._pushes_since_full >= 32:
            self._pushes_since_full = 0
            with torch.cuda.stream(self.stream):
                eng.pack_model(self.model, self.dev)                   # no-op while the parameter signatures stand
            self._rewatch()
        if self._signature() == self._captured_for:
            return
        need = eng.workspace_bytes(1, self.T)
        if self.ws.numel() < need:
            self.stream.synchronize()
            self.ws = torch.empty(need, dtype=torch.uint8, device=self.dev)
        if self.graph is not None:
            with torch.cuda.stream(self.stream):
                self._capture()
        else:
            self._captured_for = self._signature()

    def _capture(self):
        """Queue-free capture of one step with the session's own workspace; remembers which packed blob it was captured
        against (a re-pack or an adopted blob invalidates the graph: push() re-captures)."""
        eng = self.model._engine
        self.graph = torch.cuda.CUDAGraph()
        with eng.use_workspace(self.ws), torch.cuda.graph(self.graph, stream=self.stream):
            self._step()
        self._captured_for = self._signature()

    @torch.no_grad()
    def push(self, feature):
        """feature: [2048] (host tensor / numpy / device tensor).  Blocks until the newest frame's outputs are in the pinned
        result tensors and returns them."""
        eng = self.model._engine
        if torch.is_tensor(feature) and feature.is_cuda:
            feature = feature.cpu()
        self.feat_host.copy_(torch.as_tensor(feature, dtype=torch.float32).reshape(2048))
        # the blob the graph was captured against may have been re-packed (a weight change, here or by another caller: kernel selection may
        # differ, e.g. the fp16-range fallback) or replaced (adopt_blob, device move)
        self._refresh()
        for attempt in (0, 1):
            with torch.cuda.stream(self.stream), eng.use_workspace(self.ws):
                if self.graph is not None:
                    self.graph.replay()
                else:
                    with eng.lazy_status():
                        self._step()
                self.done.record(self.stream)
            self.done.synchronize()
            if eng.lib.tepose_status_peek(eng.handle) != _lib.E_TIMEOUT:
                break
            if attempt:
                _lib.check(_lib.E_TIMEOUT, 'StreamSession.push')
            # a persistent kernel gave up inside the replayed graph: clear, switch kernels, re-capture, recompute this frame
            eng.lib.tepose_status(eng.handle, self.stream.cuda_stream)
            eng._raise_if_kernel_fault('StreamSession.push')             # code 4: not a residency problem -- no switch, no re-run
            eng._degrade('StreamSession recomputes this frame and re-captures its graph')
            with torch.cuda.stream(self.stream):
                self.win.copy_(self.saved)
                if self.use_graph:
                    self._capture()
            self.stream.synchronize()
        self.frames += 1
        return self.out_host
