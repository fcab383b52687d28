"""Drop-in for the reference's lib/models/vibe.py: the VIBE temporal encoder + regressor that
bootstraps the first seqlen-1 frames (evaluate.py:89-107,233-245; demo.py:104-130,229-237).
Same constructor / forward signatures and state-dict keys; compute in libtepose_hip.so.  Every
constructor configuration of vibe.py:27-47 runs on the device: uni- or bidirectional GRU, with the
Linear (bidirectional or add_linear) or without it, residual when the output is 2048 wide."""
import os

import torch
import torch.nn as nn

from .engine import Engine, on_device, regroup_outputs
from .spin import Regressor, warm_start_from_spin


class TemporalEncoder(nn.Module):
    def __init__(self, n_layers=1, hidden_size=2048, add_linear=False, bidirectional=False, use_residual=True,
                 _engine=None):
        super().__init__()
        self.gru = nn.GRU(input_size=2048, hidden_size=hidden_size, bidirectional=bidirectional, num_layers=n_layers)
        self.linear = None                                                   # vibe.py:43-47
        if bidirectional:
            self.linear = nn.Linear(hidden_size * 2, 2048)
        elif add_linear:
            self.linear = nn.Linear(hidden_size, 2048)
        self.use_residual = use_residual
        self.n_layers, self.hidden_size = n_layers, hidden_size
        object.__setattr__(self, '_engine', _engine if _engine is not None else
                           Engine(n_layers, hidden_size, 'vibe', bidirectional=bidirectional, add_linear=add_linear))

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
        return y.view(n, t, -1)


class VIBE(nn.Module):
    def __init__(self, seqlen, batch_size=64, n_layers=1, hidden_size=2048, add_linear=False, bidirectional=False,
                 use_residual=True, pretrained='', smpl=None, smpl_mean_params=None):
        super().__init__()
        self.seqlen = seqlen
        self.batch_size = batch_size
        engine = Engine(n_layers, hidden_size, 'vibe', bidirectional=bidirectional, add_linear=add_linear)
        object.__setattr__(self, '_engine', engine)
        self.encoder = TemporalEncoder(n_layers=n_layers, hidden_size=hidden_size, bidirectional=bidirectional,
                                       add_linear=add_linear, use_residual=use_residual, _engine=engine)
        kw = {} if smpl_mean_params is None else {'smpl_mean_params': smpl_mean_params}
        self.regressor = Regressor(smpl=smpl, _engine=engine, **kw)
        warm_start_from_spin(self.regressor, pretrained)

    def forward(self, input, J_regressor=None):
        n_clips, n_frames = input.shape[:2]
        feature = self.encoder(input)
        per_frame = feature.reshape(-1, feature.size(-1))                   # vibe.py:110
        return [regroup_outputs(o, (n_clips, n_frames)) for o in self.regressor(per_frame, J_regressor=J_regressor)]
