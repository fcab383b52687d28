"""Drop-in for the reference's lib/models/tepose.py: `TemporalEncoder` and `TePose` with the
same constructor / forward signatures and state-dict keys.

The nn.GRU / nn.Linear members exist only as parameter containers (so `load_state_dict`,
`.to()`, `.parameters()` behave as in the reference); they are never called.  The forward
re-packs weights into the HIP library's layout whenever a parameter changed, then makes
one C-ABI call (tepose_forward) on the current stream.
"""
import os
import os.path as osp

import torch
import torch.nn as nn

from .engine import Engine, check_input, on_device, regroup_outputs, warn_if_training
from .smpl import BASE_DATA_DIR
from .spin import Regressor, warm_start_from_spin


class TemporalEncoder(nn.Module):
    def __init__(self, n_layers=1, seq_len=16, hidden_size=2048, _engine=None):
        super().__init__()
        self.gru_fwd = nn.GRU(input_size=2133, hidden_size=hidden_size, bidirectional=False,
                              num_layers=n_layers)
        self.gru_rec = nn.GRU(input_size=2133, hidden_size=hidden_size, bidirectional=True,
                              num_layers=n_layers)
        self.mid_frame = int(seq_len / 2)
        self.hidden_size = hidden_size
        self.n_layers = n_layers
        self.linear_fwd = nn.Linear(hidden_size, 2048)
        self.linear_rec = nn.Linear(hidden_size * 2, 2048)
        object.__setattr__(self, '_engine', _engine if _engine is not None else Engine(n_layers, hidden_size))

    def forward(self, x, is_train=False):
        warn_if_training(self, x)
        x = check_input(x)
        with on_device(x.device):
            self._engine.pack_encoder(self, x.device)
            return self._engine.encoder_fwd(x, is_train)


class TePose(nn.Module):
    def __init__(self, seqlen, batch_size=64, n_layers=1, hidden_size=2048,
                 pretrained=osp.join(BASE_DATA_DIR, 'spin_model_checkpoint.pth.tar'),
                 smpl=None, smpl_mean_params=None):
        super().__init__()
        self.seqlen = seqlen
        self.batch_size = batch_size
        engine = Engine(n_layers, hidden_size)
        object.__setattr__(self, '_engine', engine)
        self.encoder = TemporalEncoder(seq_len=seqlen, n_layers=n_layers, hidden_size=hidden_size,
                                       _engine=engine)
        kw = {} if smpl_mean_params is None else {'smpl_mean_params': smpl_mean_params}
        self.regressor = Regressor(smpl=smpl, _engine=engine, **kw)
        warm_start_from_spin(self.regressor, pretrained)

    def forward(self, input, is_train=False, J_regressor=None):
        warn_if_training(self, input)
        x = check_input(input)
        batch_size = x.shape[0]
        if batch_size == 0 or x.shape[1] == 0:
            raise ValueError('empty batch / zero-length window: torch.nn.GRU in the reference rejects it too')
        eng = self._engine
        with on_device(x.device):
            eng.pack_encoder(self.encoder, x.device)
            eng.pack_regressor(self.regressor, x.device)
            if not is_train:
                return [eng.forward(x, J_regressor)]
            feature = eng.encoder_fwd(x, True).reshape(-1, 2048)
            out = eng.regressor_fwd(feature, 3, None, ws_hint=eng.workspace(batch_size, x.shape[1], x.device))
        # two predictions per window: y_fwd and y_rec (lib/models/tepose.py:138-145)
        return [regroup_outputs(out, (batch_size, 2))]
