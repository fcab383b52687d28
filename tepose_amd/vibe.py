"""Drop-in for the reference's lib/models/vibe.py: the VIBE temporal encoder + regressor that
bootstraps the first seqlen-1 frames (evaluate.py:89-107,233-245; demo.py:104-130,229-237).
Same constructor / forward signatures and state-dict keys; compute in libtepose_hip.so.
Only the configuration the TePose callers use is accelerated: uni-directional GRU with
add_linear=True (evaluate.py:93-101)."""
import os

import torch
import torch.nn as nn

from .engine import Engine, on_device, regroup_outputs
from .spin import Regressor, warm_start_from_spin


class TemporalEncoder(nn.Module):
    def __init__(self, n_layers=1, hidden_size=2048, add_linear=False, bidirectional=False, use_residual=True,
                 _engine=None):
        super().__init__()
        if bidirectional or not add_linear:
            raise NotImplementedError('tepose_amd.vibe: only bidirectional=False, add_linear=True (the '
                                      'configuration of evaluate.py:93-101 / demo.py:107-115) is implemented')
        self.gru = nn.GRU(input_size=2048, hidden_size=hidden_size, bidirectional=False, num_layers=n_layers)
        self.linear = nn.Linear(hidden_size, 2048)
        self.use_residual = use_residual
        self.n_layers, self.hidden_size = n_layers, hidden_size
        object.__setattr__(self, '_engine', _engine if _engine is not None else Engine(n_layers, hidden_size, 'vibe'))

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError('tepose_amd runs on MI355X only: move the model and input to a cuda device')
        n, t, f = x.shape
        if f != 2048:
            raise ValueError('VIBE encoder input must be [N, T, 2048]')
        x = x.float().contiguous()
        with on_device(x.device):
            self._engine.pack_vibe_encoder(self, x.device)
            y = self._engine.vibe_encoder_fwd(x, self.use_residual)
        return y.view(n, t, 2048)


class VIBE(nn.Module):
    def __init__(self, seqlen, batch_size=64, n_layers=1, hidden_size=2048, add_linear=False, bidirectional=False,
                 use_residual=True, pretrained='', smpl=None, smpl_mean_params=None):
        super().__init__()
        self.seqlen = seqlen
        self.batch_size = batch_size
        engine = Engine(n_layers, hidden_size, 'vibe')
        object.__setattr__(self, '_engine', engine)
        self.encoder = TemporalEncoder(n_layers=n_layers, hidden_size=hidden_size, bidirectional=bidirectional,
                                       add_linear=add_linear, use_residual=use_residual, _engine=engine)
        kw = {} if smpl_mean_params is None else {'smpl_mean_params': smpl_mean_params}
        self.regressor = Regressor(smpl=smpl, _engine=engine, **kw)
        warm_start_from_spin(self.regressor, pretrained)

    def forward(self, input, J_regressor=None):
        n_clips, n_frames = input.shape[:2]
        per_frame = self.encoder(input).reshape(n_clips * n_frames, 2048)
        return [regroup_outputs(o, (n_clips, n_frames)) for o in self.regressor(per_frame, J_regressor=J_regressor)]
