"""Host-side owner of one tepose_model handle: the packed-weight blob, the workspace,
and the calls into the C ABI.  PyTorch is used for device memory and streams only.
"""
import contextlib
import ctypes
import os
import warnings
from ctypes import c_double, c_int, c_int32, c_void_p

import torch

from . import _lib
from .synth import NUM_VERTS


def _sig(tensors):
    # (address, in-place version) of every tensor: `.to()` / `load_state_dict` / an optimiser step change one of the two
    # (the address also identifies the device); ~0.2 us per tensor, checked on every forward
    return tuple([(t.data_ptr(), t._version) for t in tensors])


class on_device:
    """`with torch.cuda.device(d)` only when d is not the current device already (the context manager costs ~10 us per
    forward, comparable to a launch-bound B = 1 forward's whole host budget)."""
    __slots__ = ('ctx',)

    def __init__(self, device):
        idx = device.index if device.index is not None else torch.cuda.current_device()
        self.ctx = None if idx == torch.cuda.current_device() else torch.cuda.device(idx)

    def __enter__(self):
        if self.ctx is not None:
            self.ctx.__enter__()

    def __exit__(self, *a):
        if self.ctx is not None:
            return self.ctx.__exit__(*a)
        return False


def _dev_f32(t, device):
    """fp32 contiguous copy/view of t on device (no copy when already there)."""
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


class Engine:
    """One per TePose / standalone TemporalEncoder / standalone Regressor."""

    def __init__(self, n_layers, hidden, kind='tepose', bidirectional=False, add_linear=True):
        self.lib = _lib.load()
        h = c_void_p()
        self.kind = kind
        if kind == 'vibe':
            _lib.check(self.lib.tepose_create_vibe_ex(int(n_layers), int(hidden), int(bool(bidirectional)), int(bool(add_linear)),
                                                      ctypes.byref(h)), 'tepose_create_vibe_ex')
        else:
            _lib.check(self.lib.tepose_create(int(n_layers), int(hidden), ctypes.byref(h)), 'tepose_create')
        self.handle = h
        self.n_layers, self.hidden = int(n_layers), int(hidden)
        self.blob = None
        self.device = None
        self._sig_enc = self._sig_reg = self._sig_smpl = None
        self._ws = None
        self._ws_private = None        # a caller-owned workspace (use_workspace): StreamSession's captured graphs
        self._ws_need = {}
        self.packed_generation = 0     # bumped by every (re)pack: the workspace size can depend on what was packed
        self._jreg_cache = {}
        self.profiling = False
        # What happens when a persistent small-batch kernel gives up (TEPOSE_E_TIMEOUT, include/tepose_amd.h):
        #   'sync' (default): every forward that may have launched one synchronises its stream and reads the handle's fault
        #          word before returning -- on a fault the handle switches to the step-per-launch kernels, warns, and the
        #          forward is re-run, so the caller never sees the NaN outputs (the reference's callers .cpu() the outputs
        #          right after the forward, evaluate.py:258-261: the sync is theirs anyway);
        #   'lazy': no sync; the fault surfaces as TeposeTimeout at the next call on the handle or at check_status() --
        #          for callers that own their sync points (tepose_amd.driver, bench.py).
        self.status_mode = os.environ.get('TEPOSE_STATUS_CHECK', 'sync')
        if self.status_mode not in ('sync', 'lazy'):
            raise ValueError("TEPOSE_STATUS_CHECK must be 'sync' or 'lazy'")
        self._uses_persistent = {}
        self.degraded = False          # a fault switched this handle to the step-per-launch kernels

    def __del__(self):
        try:
            if getattr(self, 'handle', None):
                self.lib.tepose_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    # ------------------------------------------------------------------ blob / packing
    @property
    def packed_bytes(self):
        return int(self.lib.tepose_packed_bytes(self.handle))

    def _ensure_blob(self, device):
        if self.blob is None or self.device != device:
            self.blob = torch.zeros(self.packed_bytes, dtype=torch.uint8, device=device)
            self.device = device
            _lib.check(self.lib.tepose_set_blob(self.handle, self.blob.data_ptr(), self.blob.numel()),
                       'tepose_set_blob')
            self._sig_enc = self._sig_reg = self._sig_smpl = None
            self._ws = None
            self._jreg_cache = {}

    def fp32_ranges(self):
        """(offset, size) byte ranges of the blob that a sender has to transmit: the fp32 sections and the header; every
        hi / lo plane is re-derived from them by the receiver (adopt_blob(..., derive=True))."""
        from ctypes import c_size_t
        off, size = (c_size_t * 8)(), (c_size_t * 8)()
        n = self.lib.tepose_fp32_ranges(self.handle, off, size, 8)
        if n < 0:
            _lib.check(n, 'tepose_fp32_ranges')
        return [(int(off[i]), int(size[i])) for i in range(n)]

    def adopt_blob(self, blob, model, derive=False):
        """Use a packed blob produced by another Engine of the same (n_layers, hidden) --
        e.g. received through an RCCL broadcast from rank 0 -- instead of packing `model`'s
        own parameters, which are then ignored until one of them changes.  derive=True: only the
        fp32_ranges() of `blob` are valid (the rest zero); the planes are rebuilt here."""
        assert blob.dtype == torch.uint8 and blob.numel() >= self.packed_bytes
        self.blob, self.device = blob, blob.device
        _lib.check(self.lib.tepose_set_blob(self.handle, blob.data_ptr(), blob.numel()), 'tepose_set_blob')
        if derive:
            _lib.check(self.lib.tepose_derive_planes(self.handle, self._stream()), 'tepose_derive_planes')
        else:
            _lib.check(self.lib.tepose_adopt_blob(self.handle), 'tepose_adopt_blob')
        self.packed_generation += 1
        self._sig_enc = _sig(self._vibe_enc_tensors(model.encoder) if self.kind == 'vibe' else self._enc_tensors(model.encoder))
        self._sig_reg = _sig(self._reg_tensors(model.regressor))
        self._sig_smpl = (id(model.regressor.smpl),) + _sig(self._smpl_tensors(model.regressor.smpl))
        self._ws = None
        self._jreg_cache = {}

    def _enc_tensors(self, enc):
        names = getattr(self, '_enc_names', None)
        if names is None:                      # parameter names in the C ABI's order, built once
            fwd = ['%s_l%d' % (k, l) for l in range(self.n_layers) for k in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh')]
            rec = ['%s_l%d%s' % (k, l, sfx) for l in range(self.n_layers) for sfx in ('', '_reverse')
                   for k in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh')]
            names = self._enc_names = (fwd, rec)
        gf, gr = enc.gru_fwd._parameters, enc.gru_rec._parameters
        return [gf[n] for n in names[0]] + [gr[n] for n in names[1]] + \
            [enc.linear_fwd.weight, enc.linear_fwd.bias, enc.linear_rec.weight, enc.linear_rec.bias]

    @staticmethod
    def _reg_tensors(reg):
        return [reg.fc1.weight, reg.fc1.bias, reg.fc2.weight, reg.fc2.bias, reg.decpose.weight, reg.decpose.bias,
                reg.decshape.weight, reg.decshape.bias, reg.deccam.weight, reg.deccam.bias,
                reg.init_pose, reg.init_shape, reg.init_cam]

    @staticmethod
    def _smpl_tensors(smpl):
        return [smpl.v_template, smpl.shapedirs, smpl.posedirs, smpl.J_regressor, smpl.lbs_weights,
                smpl.J_regressor_extra, smpl.parents]

    @staticmethod
    def _stream():
        return torch.cuda.current_stream().cuda_stream

    def pack_encoder(self, enc, device):
        """enc: module with gru_fwd, gru_rec (nn.GRU) and linear_fwd, linear_rec (nn.Linear)."""
        ts = self._enc_tensors(enc)
        sig = _sig(ts)
        if sig == self._sig_enc and self.device == device:
            return
        self._ensure_blob(device)
        keep = [_dev_f32(t, device) for t in ts]
        arr = _lib.ptr_array([t.data_ptr() for t in keep])
        _lib.check(self.lib.tepose_pack_encoder(self.handle, arr, len(keep), self._stream()), 'tepose_pack_encoder')
        self._sig_enc = sig
        self.packed_generation += 1

    def pack_regressor(self, reg, device):
        self._ensure_blob(device)
        ts = self._reg_tensors(reg)
        sig = _sig(ts)
        if sig != self._sig_reg:
            keep = [_dev_f32(t, device) for t in ts]
            arr = _lib.ptr_array([t.data_ptr() for t in keep])
            _lib.check(self.lib.tepose_pack_regressor(self.handle, arr, len(keep), self._stream()),
                       'tepose_pack_regressor')
            self._sig_reg = sig
            self.packed_generation += 1
        self.pack_smpl(reg.smpl, device)

    def pack_smpl(self, smpl, device):
        self._ensure_blob(device)
        ts = self._smpl_tensors(smpl)
        sig = (id(smpl),) + _sig(ts)
        if sig != self._sig_smpl:
            keep = [_dev_f32(t, device) for t in ts[:6]]
            if tuple(keep[0].shape) != (NUM_VERTS, 3) or tuple(keep[2].shape) != (207, NUM_VERTS * 3):
                raise _lib.TeposeError('SMPL tables have unexpected shapes')
            par = (c_int32 * 24)(*[int(p) for p in smpl.parents.detach().cpu().tolist()])
            _lib.check(self.lib.tepose_pack_smpl(self.handle, *[t.data_ptr() for t in keep], par, self._stream()),
                       'tepose_pack_smpl')
            self._sig_smpl = sig
            self.packed_generation += 1

    def jreg(self, J, device):
        """Packed CSR of an evaluation joint regressor [17,6890] (any device; cached)."""
        if J is None:
            return None, 0
        key = (J.data_ptr(), J._version, J.device.type)
        hit = self._jreg_cache.get(key)
        if hit is None:
            if tuple(J.shape) != (17, NUM_VERTS):
                raise _lib.TeposeError('J_regressor must be [17, 6890], got %s' % (tuple(J.shape),))
            src = _dev_f32(J, device)
            buf = torch.zeros(int(self.lib.tepose_jreg_packed_bytes()), dtype=torch.uint8, device=device)
            _lib.check(self.lib.tepose_pack_jreg(src.data_ptr(), buf.data_ptr(), self._stream()), 'tepose_pack_jreg')
            if len(self._jreg_cache) > 8:
                self._jreg_cache.clear()
            hit = self._jreg_cache[key] = (buf, J)   # keep J alive so its data_ptr stays unique
        return hit[0], hit[0].data_ptr()

    # ------------------------------------------------------------------ fault channel of the persistent kernels
    def uses_persistent(self, B):
        k = (int(B), self.packed_generation, self.degraded)
        v = self._uses_persistent.get(k)
        if v is None:
            if len(self._uses_persistent) > 64:
                self._uses_persistent.clear()
            v = self._uses_persistent[k] = bool(self.lib.tepose_uses_persistent(self.handle, int(B), 0))
        return v

    def _degrade(self, what):
        """A bounded wait expired: from now on this handle runs the step-per-launch HIP kernels (same results, no
        residency requirement)."""
        _lib.check(self.lib.tepose_set_persistent(self.handle, 0), 'tepose_set_persistent')
        self.degraded = True
        warnings.warn('tepose_amd: a persistent small-batch kernel gave up waiting for its peer workgroups (%s) -- is the GPU '
                      'shared with another process or CU-masked?  This model now uses the step-per-launch kernels FOR THE REST OF '
                      'ITS LIFE (set TEPOSE_PERSISTENT=0 to start that way; Engine.set_persistent(True) -- '
                      'model._engine.set_persistent(True) -- switches back once the GPU is yours again).' % what,
                      RuntimeWarning, stacklevel=4)

    def set_persistent(self, on):
        """Select the persistent small-batch kernels (True: the default) or the step-per-launch kernels (False).  A fault
        switches a handle to the latter for good (`degraded`); this is the way back after a transient contention.  Graphs
        captured with `torch.cuda.graph` keep the kernels they were captured with: re-capture after a switch."""
        _lib.check(self.lib.tepose_set_persistent(self.handle, 1 if on else 0), 'tepose_set_persistent')
        self.degraded = False if on else self.degraded
        self._uses_persistent.clear()

    def _raise_if_kernel_fault(self, what):
        """Fault code 4 = the barrier-free projection kernel gave up a bounded LDS poll (a kernel bug or a hardware fault): nothing the step-per-launch
        kernels would cure, so no switch, no shared-GPU warning, no silent re-run -- a distinct exception."""
        if int(self.lib.tepose_fault_code(self.handle)) == 4:
            raise _lib.TeposeKernelFault('%s: a wave of gemm_h3s_persist16c_kernel gave up a bounded LDS poll (fault code 4) -- the outputs of that forward '
                                         'are invalid; this is not a residency problem (tepose_debug_kernel_errors() = %d)'
                                         % (what, int(self.lib.tepose_debug_kernel_errors())))

    def check_status(self):
        """Synchronise the current stream and raise TeposeTimeout if a forward on this handle gave up since the last
        check (its outputs are NaN).  The handle is switched to the step-per-launch kernels first, so a re-run works."""
        rc = self.lib.tepose_status(self.handle, self._stream())
        if rc == _lib.E_TIMEOUT:
            self._raise_if_kernel_fault('tepose_status')
            self._degrade('outputs since the last check are invalid')
        _lib.check(rc, 'tepose_status')

    @contextlib.contextmanager
    def lazy_status(self):
        old, self.status_mode = self.status_mode, 'lazy'
        try:
            yield self
        finally:
            self.status_mode = old

    def _forward_status(self, ws):
        """Did the forward that just ran with workspace `ws` give up?  Per forward (the status words live in its workspace), so
        the answer does not depend on what other streams / handles-sharing threads did in the meantime."""
        if ws is None:
            return self.lib.tepose_status(self.handle, self._stream())
        return self.lib.tepose_forward_status(self.handle, ws.data_ptr(), self._stream())

    def _run(self, B, call, ws=None):
        """call() queues one forward on the current stream and returns its outputs.  (One Engine = one workspace: a model is
        driven by one stream / thread at a time; concurrent streams take one model each, or the C ABI with one workspace each.)
        sync mode synchronises the stream of every small-batch forward; a forward captured into a hipGraph is NOT checked at
        replay -- call check_status() after replays, and re-capture after a fault switched the kernels."""
        try:
            out = call()
        except _lib.TeposeTimeout:
            # refused up front: an EARLIER forward on this handle gave up (lazy mode); clear, switch kernels, tell the caller
            self.lib.tepose_status(self.handle, self._stream())
            self._raise_if_kernel_fault('an earlier forward')
            self._degrade('outputs of earlier forwards are invalid')
            raise
        if self.status_mode == 'sync' and not torch.cuda.is_current_stream_capturing():
            if self.uses_persistent(B):
                rc = self._forward_status(ws)
                if rc == _lib.E_TIMEOUT:
                    self._raise_if_kernel_fault('this forward')
                    self._degrade('re-running this forward')
                    out = call()
                    rc = self._forward_status(ws)
                _lib.check(rc, 'tepose_forward_status')
            elif self.lib.tepose_status_peek(self.handle) == _lib.E_TIMEOUT:
                # large batch: not synchronised by the library, but a fault word raised by a forward that has ALREADY finished (one host-memory read)
                # is reported now rather than at the next call
                # (handle-wide: the forward that gave up may be another caller's -- its own tepose_forward_status still reports it, the library
                # tracks collections per forward -- but the handle is switched like on every other path, so the raise is never the only effect)
                self.lib.tepose_status(self.handle, self._stream())
                self._raise_if_kernel_fault('a forward on this handle')
                self._degrade('a small-batch forward on this handle gave up; its outputs are invalid')
                _lib.check(_lib.E_TIMEOUT, 'tepose_status')
        return out

    # ------------------------------------------------------------------ forward
    def workspace(self, B, T, device):
        k = (int(B), int(T), self.packed_generation)
        need = self._ws_need.get(k)
        if need is None:
            need = self._ws_need[k] = int(self.lib.tepose_workspace_bytes(self.handle, int(B), int(T)))
            if len(self._ws_need) > 64:
                self._ws_need = {k: need}
        if self._ws_private is not None:
            if self._ws_private.numel() < need or self._ws_private.device != device:
                raise RuntimeError('the caller-owned workspace (%d bytes on %s) is too small for B = %d, T = %d (%d bytes on %s)'
                                   % (self._ws_private.numel(), self._ws_private.device, B, T, need, device))
            return self._ws_private
        if self._ws is None or self._ws.numel() < need or self._ws.device != device:
            self._ws = None
            self._ws = torch.empty(need, dtype=torch.uint8, device=device)
        return self._ws

    def workspace_bytes(self, B, T):
        return int(self.lib.tepose_workspace_bytes(self.handle, int(B), int(T)))

    @contextlib.contextmanager
    def use_workspace(self, ws):
        """Every forward queued inside the block uses the caller's tensor `ws` instead of the engine's shared, growable
        workspace.  For callers whose kernels outlive the call -- a captured hipGraph bakes the workspace ADDRESS into its
        kernel arguments, and the shared workspace is freed and re-allocated whenever a later call needs more bytes or the
        blob is re-packed (replays into freed memory)."""
        old, self._ws_private = self._ws_private, ws
        try:
            yield self
        finally:
            self._ws_private = old

    def encoder_fwd(self, x, is_train):
        B, T = x.shape[:2]
        ws = self.workspace(B, T, x.device)

        def call():
            feat = torch.empty((B, 2, 2048) if is_train else (B, 2048), dtype=torch.float32, device=x.device)
            _lib.check(self.lib.tepose_encoder_fwd(self.handle, x.data_ptr(), B, T, 1 if is_train else 0,
                                                   feat.data_ptr(), ws.data_ptr(), ws.numel(), self._stream()),
                       'tepose_encoder_fwd')
            return feat
        return self._run(B, call, ws)

    def regressor_fwd(self, feat, n_iter, J_regressor, ws_hint=None, init=(None, None, None)):
        N = feat.shape[0]
        dev = feat.device
        init = [None if t is None else _dev_f32(t, dev) for t in init]      # per-call [N,144] / [N,10] / [N,3]
        for t, w in zip(init, (144, 10, 3)):
            if t is not None and tuple(t.shape) != (N, w):
                raise ValueError('init_pose / init_shape / init_cam must be [N,144] / [N,10] / [N,3] '
                                 '(lib/models/spin.py:240-251), got %s' % (tuple(t.shape),))
        ws = self.workspace(max(1, (N + 1) // 2), 1, dev) if ws_hint is None else ws_hint
        _, jp = self.jreg(J_regressor, dev)
        nj = 14 if J_regressor is not None else 49

        def call():
            out = self._outputs(N, nj, dev)
            _lib.check(self.lib.tepose_regressor_fwd_init(
                self.handle, feat.data_ptr(), N, int(n_iter), *[None if t is None else t.data_ptr() for t in init], jp,
                out['theta'].data_ptr(), out['verts'].data_ptr(),
                out['kp_3d'].data_ptr(), out['kp_2d'].data_ptr(), out['rotmat'].data_ptr(), ws.data_ptr(), ws.numel(),
                self._stream()), 'tepose_regressor_fwd_init')
            return out
        return self._run(N, call, ws)

    @staticmethod
    def _outputs(N, nj, dev):
        """The five output tensors of N rows: fresh, contiguous, independently freeable (as the reference's are)."""
        return {
            'theta': torch.empty((N, 85), dtype=torch.float32, device=dev),
            'verts': torch.empty((N, NUM_VERTS, 3), dtype=torch.float32, device=dev),
            'kp_2d': torch.empty((N, nj, 2), dtype=torch.float32, device=dev),
            'kp_3d': torch.empty((N, nj, 3), dtype=torch.float32, device=dev),
            'rotmat': torch.empty((N, 24, 3, 3), dtype=torch.float32, device=dev),
        }

    def _out_views(self, out, B, nj, dev):
        """The five output tensors of a forward: fresh ones, or the caller's (`out`: dict key -> contiguous fp32 cuda tensor whose first B rows are
        written -- the clip drivers hand in slices of their result buffers, so that a window step costs no copy); keys the caller leaves out are fresh."""
        if out is None:
            return self._outputs(B, nj, dev)
        shapes = {'theta': (85,), 'verts': (NUM_VERTS, 3), 'kp_2d': (nj, 2), 'kp_3d': (nj, 3), 'rotmat': (24, 3, 3)}
        res = {}
        for k, tail in shapes.items():
            t = out.get(k)
            if t is None:
                t = torch.empty((B,) + tail, dtype=torch.float32, device=dev)
            elif not (t.is_cuda and t.device == dev and t.dtype == torch.float32 and t.is_contiguous() and t.shape[0] >= B and tuple(t.shape[1:]) == tail):
                raise ValueError('out[%r] must be a contiguous fp32 tensor on %s of shape [>= %d, %s], got %s %s' % (k, dev, B, tail, tuple(t.shape), t.dtype))
            res[k] = t[:B]
        return res

    def forward(self, x, J_regressor, out=None):
        B, T = x.shape[:2]
        dev = x.device
        ws = self.workspace(B, T, dev)
        _, jp = self.jreg(J_regressor, dev)
        nj = 14 if J_regressor is not None else 49
        given = out

        def call():
            out = self._out_views(given, B, nj, dev)
            _lib.check(self.lib.tepose_forward(
                self.handle, x.data_ptr(), B, T, jp, out['theta'].data_ptr(), out['verts'].data_ptr(),
                out['kp_3d'].data_ptr(), out['kp_2d'].data_ptr(), out['rotmat'].data_ptr(), ws.data_ptr(), ws.numel(),
                self._stream()), 'tepose_forward')
            return out
        return self._run(B, call, ws)

    def smpl_fwd(self, pose, betas, pose2rot):
        """pose [N,72] (axis-angle) or [N,24,3,3]; betas [N,10] -> verts [N,6890,3], joints [N,49,3]."""
        N, dev = pose.shape[0], pose.device
        ws = self.workspace(max(1, (N + 1) // 2), 1, dev)
        verts = torch.empty((N, NUM_VERTS, 3), dtype=torch.float32, device=dev)
        joints = torch.empty((N, 49, 3), dtype=torch.float32, device=dev)
        _lib.check(self.lib.tepose_smpl_fwd(self.handle, 1 if pose2rot else 0, pose.data_ptr(), betas.data_ptr(), N,
                                            verts.data_ptr(), joints.data_ptr(), ws.data_ptr(), ws.numel(),
                                            self._stream()), 'tepose_smpl_fwd')
        return verts, joints

    def joints_from_verts(self, verts, J_regressor):
        """verts [N,6890,3] (device) -> the 14 LSP joints of the H36M regressor [N,14,3] (evaluate.py:289-291)."""
        N, dev = verts.shape[0], verts.device
        _, jp = self.jreg(J_regressor, dev)
        out = torch.empty((N, 14, 3), dtype=torch.float32, device=dev)
        _lib.check(self.lib.tepose_joints_from_verts(self.handle, jp, verts.data_ptr(), N, out.data_ptr(), self._stream()), 'tepose_joints_from_verts')
        return out

    # ------------------------------------------------------------------ cached layer-0 projections
    @property
    def gate_width(self):
        """Floats per frame of cached layer-0 gate pre-activations (3 directions x 3 gates x Hp)."""
        return 9 * ((self.hidden + 63) // 64 * 64)

    def project_frames(self, feat_ptr, feat_ld, theta_ptr, theta_ld, B, out_ptr, out_ld, ws):
        _lib.check(self.lib.tepose_project_frames(self.handle, feat_ptr, feat_ld, theta_ptr, theta_ld, B, out_ptr,
                                                  out_ld, ws.data_ptr(), ws.numel(), self._stream()),
                   'tepose_project_frames')

    def project_frame_pair(self, feat_prev_ptr, feat_new_ptr, feat_ld, theta_prev_ptr, theta_ld, B, out_prev_ptr, out_prev_ld, out_new_ptr, out_new_ld, ws):
        _lib.check(self.lib.tepose_project_frame_pair(self.handle, feat_prev_ptr, feat_new_ptr, feat_ld, theta_prev_ptr, theta_ld, B, out_prev_ptr,
                                                      out_prev_ld, out_new_ptr, out_new_ld, ws.data_ptr(), ws.numel(), self._stream()),
                   'tepose_project_frame_pair')

    def forward_cached(self, ring, first_slot, newest, B, T, J_regressor, out=None):
        """ring [C, R, 9Hp], newest [C, 9Hp] (rows [0,B) used) -> output dict like forward(); `out`: see _out_views."""
        dev = ring.device
        ws = self.workspace(B, T, dev)
        _, jp = self.jreg(J_regressor, dev)
        nj = 14 if J_regressor is not None else 49
        given = out

        def call():
            out = self._out_views(given, B, nj, dev)
            _lib.check(self.lib.tepose_forward_cached(
                self.handle, ring.data_ptr(), ring.shape[1], int(first_slot), ring.stride(0), newest.data_ptr(),
                newest.stride(0), B, T, jp, out['theta'].data_ptr(), out['verts'].data_ptr(), out['kp_3d'].data_ptr(),
                out['kp_2d'].data_ptr(), out['rotmat'].data_ptr(), ws.data_ptr(), ws.numel(), self._stream()),
                'tepose_forward_cached')
            return out
        return self._run(B, call, ws)

    def window_step(self, feat_prev_ptr, feat_new_ptr, feat_ld, theta_prev_ptr, theta_ld, out_prev_ptr, out_prev_ld, ring, first_slot, newest, B, T, J_regressor,
                    pair_ws, out=None):
        """One lock-step of the clip driver as one library call (tepose_window_step): the step's two layer-0 projections as one product (previous newest
        frame + its theta -> `out_prev`, newest frame + zero theta -> `newest`), then the forward of the window from the cached projections."""
        dev = ring.device
        ws = self.workspace(B, T, dev)
        _, jp = self.jreg(J_regressor, dev)
        nj = 14 if J_regressor is not None else 49
        given = out

        def call():
            out = self._out_views(given, B, nj, dev)
            _lib.check(self.lib.tepose_window_step(
                self.handle, feat_prev_ptr, feat_new_ptr, feat_ld, theta_prev_ptr, theta_ld, out_prev_ptr, out_prev_ld, newest.data_ptr(), newest.stride(0),
                ring.data_ptr(), ring.shape[1], int(first_slot), ring.stride(0), B, T, jp, out['theta'].data_ptr(), out['verts'].data_ptr(),
                out['kp_3d'].data_ptr(), out['kp_2d'].data_ptr(), out['rotmat'].data_ptr(), ws.data_ptr(), ws.numel(), pair_ws.data_ptr(), pair_ws.numel(),
                self._stream()), 'tepose_window_step')
            return out
        return self._run(B, call, ws)

    # ------------------------------------------------------------------ VIBE bootstrap encoder
    def _vibe_enc_tensors(self, enc):
        ts = []
        for l in range(self.n_layers):
            for sfx in ('', '_reverse') if enc.gru.bidirectional else ('',):
                ts += [getattr(enc.gru, '%s_l%d%s' % (k, l, sfx)) for k in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh')]
        if enc.linear is not None:
            ts += [enc.linear.weight, enc.linear.bias]
        return ts

    def pack_model(self, model, device):
        """Pack every part of a TePose / VIBE module (encoder, regressor, SMPL) that changed since the last pack."""
        with on_device(device):
            if self.kind == 'vibe':
                self.pack_vibe_encoder(model.encoder, device)
            else:
                self.pack_encoder(model.encoder, device)
            self.pack_regressor(model.regressor, device)

    def pack_vibe_encoder(self, enc, device):
        ts = self._vibe_enc_tensors(enc)
        sig = _sig(ts)
        if sig == self._sig_enc and self.device == device:
            return
        self._ensure_blob(device)
        keep = [_dev_f32(t, device) for t in ts]
        arr = _lib.ptr_array([t.data_ptr() for t in keep])
        _lib.check(self.lib.tepose_pack_vibe_encoder(self.handle, arr, len(keep), self._stream()),
                   'tepose_pack_vibe_encoder')
        self._sig_enc = sig
        self.packed_generation += 1

    def vibe_encoder_fwd(self, x, use_residual):
        B, N = x.shape[:2]
        need = int(self.lib.tepose_vibe_workspace_bytes(self.handle, B, N))
        ws = torch.empty(need, dtype=torch.uint8, device=x.device)
        feat = torch.empty((B * N, int(self.lib.tepose_vibe_feature_dim(self.handle))), dtype=torch.float32, device=x.device)
        _lib.check(self.lib.tepose_vibe_encoder_fwd(self.handle, x.data_ptr(), B, N, 1 if use_residual else 0,
                                                    feat.data_ptr(), ws.data_ptr(), ws.numel(), self._stream()),
                   'tepose_vibe_encoder_fwd')
        return feat

    def kernel_info(self):
        """{'projection': symbol, 'gru_step': symbol}: the kernels this handle's knobs select for the dominant launches."""
        txt = (self.lib.tepose_kernel_info(self.handle) or b'').decode()
        return dict(kv.split('=', 1) for kv in txt.split(';') if '=' in kv) if txt else {}

    def set_option(self, name, value):
        """One integer option of THIS handle (include/tepose_amd.h tepose_set_option; e.g. 'S_MIN_B', 'SKINNY_H3_MAX_M'); refused once anything is packed."""
        _lib.check(self.lib.tepose_set_option(self.handle, str(name).encode(), int(value)), 'tepose_set_option(%s)' % name)
        self._uses_persistent.clear()
        self._ws_need.clear()

    def get_option(self, name):
        return int(self.lib.tepose_get_option(self.handle, str(name).encode()))

    def select_kernels(self, B, T):
        """The kernel selection of an eval forward of B windows x T frames on this handle, family -> kernel symbol / layout (the C library's own
        dispatch function: csrc/api.hip select_kernels).  No device needed."""
        txt = (self.lib.tepose_select_kernels(self.handle, int(B), int(T)) or b'').decode()
        return dict(kv.split('=', 1) for kv in txt.split(';') if '=' in kv) if txt else {}

    # ------------------------------------------------------------------ profiling hook (bench.py)
    def profile_enable(self, on):
        _lib.check(self.lib.tepose_profile_enable(self.handle, 1 if on else 0), 'tepose_profile_enable')

    def profile_read(self):
        ms, n, fl = c_double(), c_int(), c_double()
        _lib.check(self.lib.tepose_profile_read(self.handle, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl)),
                   'tepose_profile_read')
        return ms.value, n.value, fl.value

    def profile_read_l1proj(self):
        """(ms, forwards, flops per forward) of the layer >= 1 input projections; call before profile_read_gru (which resets)."""
        ms, n, fl = c_double(), c_int(), c_double()
        _lib.check(self.lib.tepose_profile_read_l1proj(self.handle, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl)),
                   'tepose_profile_read_l1proj')
        return ms.value, n.value, fl.value

    def profile_read_gru(self):
        ms, n, fl = c_double(), c_int(), c_double()
        _lib.check(self.lib.tepose_profile_read_gru(self.handle, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl)),
                   'tepose_profile_read_gru')
        return ms.value, n.value, fl.value


_warned_train = [False]


def warn_if_training(module, *tensors):
    """The HIP path is inference-only: no autograd graph is recorded and Dropout (spin.py:262-265) is never applied.
    Outputs in train mode are therefore eval-mode outputs; say so once instead of diverging silently."""
    if _warned_train[0]:
        return
    grad = torch.is_grad_enabled() and (any(t is not None and torch.is_tensor(t) and t.requires_grad for t in tensors) or
                                        any(p.requires_grad for p in module.parameters()))
    if module.training or grad:
        _warned_train[0] = True
        warnings.warn('tepose_amd is inference-only: outputs carry no autograd graph and Dropout is not applied'
                      + (' (module is in train mode: call model.eval())' if module.training else '')
                      + (' (grad is enabled: wrap the call in torch.no_grad())' if grad else ''), RuntimeWarning,
                      stacklevel=3)


def check_input(x):
    if not torch.is_tensor(x) or x.dim() != 3 or x.shape[2] != 2133:
        raise ValueError('input must be a [B, T, 2133] tensor (reference lib/models/tepose.py:54)')
    if not x.is_cuda:
        raise RuntimeError('tepose_amd runs on MI355X only: move the model and input to a cuda device '
                           '(there is no CPU path)')
    if x.dtype != torch.float32:
        x = x.float()
    return x.contiguous()


# trailing shape of every output of the regressor, per row
_OUT_TAIL = {'theta': (-1,), 'verts': (-1, 3), 'kp_2d': (-1, 2), 'kp_3d': (-1, 3), 'rotmat': (-1, 3, 3)}


def regroup_outputs(out, lead):
    """View the regressor's per-row outputs [N, ...] as [*lead, ...] (N = prod(lead)): what the
    reference does key by key after its regressor call (lib/models/tepose.py:130-145,
    lib/models/vibe.py:110-115)."""
    for key, tail in _OUT_TAIL.items():
        out[key] = out[key].reshape(tuple(lead) + tail)
    return out
