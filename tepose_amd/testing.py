"""Helpers shared by tests, bench.py and __graft_entry__.smoke(): build a TePose with the
deterministic synthetic weights / SMPL tables of tepose_amd.synth."""
import torch

from . import synth
from .smpl import SMPL
from .tepose import TePose


def build_model(n_layers=2, hidden=1024, seed=0, device='cuda', smpl_np=None, state=None, seqlen=16):
    smpl_np = synth.synthetic_smpl(0) if smpl_np is None else smpl_np
    state = synth.synthetic_state_dict(n_layers, hidden, seed) if state is None else state
    smpl = SMPL.from_tables(smpl_np)
    mean = {'pose': state['regressor.init_pose'][0], 'shape': state['regressor.init_shape'][0],
            'cam': state['regressor.init_cam'][0]}
    model = TePose(seqlen=seqlen, n_layers=n_layers, hidden_size=hidden, pretrained='', smpl=smpl,
                   smpl_mean_params=mean)
    sd = model.state_dict()
    for k, v in state.items():
        assert k in sd and tuple(sd[k].shape) == tuple(v.shape), k
        sd[k] = torch.from_numpy(v)
    model.load_state_dict(sd, strict=True)
    return model.to(device).eval(), state, smpl_np
