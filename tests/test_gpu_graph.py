"""GPU: a whole forward captured into a hipGraph (torch.cuda.CUDAGraph) and replayed.  The C ABI's forward entry points
neither synchronise nor allocate (include/tepose_amd.h), the per-forward arrival counters are cleared by the forward's own
first kernel / memset node, and the persistent kernels count epochs within the call -- so a captured forward must replay
to the same bits, for the launch-bound small batches (persistent kernels) and for a mid-size batch."""
import pytest
import torch

from tepose_amd import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('L,H,B,T', [(2, 1024, 1, 16), (2, 1024, 3, 6), (2, 1024, 40, 6), (2, 256, 300, 4), (1, 64, 2, 5)])
def test_captured_forward_replays_bit_identically(L, H, B, T):
    from tepose_amd.testing import build_model
    smpl_np = synth.synthetic_smpl(0)
    model, _, _ = build_model(L, H, seed=5, device='cuda', smpl_np=smpl_np)
    J = torch.from_numpy(smpl_np['J_regressor_h36m'])
    x = torch.from_numpy(synth.synthetic_windows(B, T, 3)).cuda()
    x2 = torch.from_numpy(synth.synthetic_windows(B, T, 4)).cuda()
    with torch.no_grad():
        eager = {k: v.clone() for k, v in model(x, J_regressor=J)[0].items()}          # also packs, sizes the workspace
        eager2 = {k: v.clone() for k, v in model(x2, J_regressor=J)[0].items()}
        static_x = x.clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                model(static_x, J_regressor=J)
        torch.cuda.current_stream().wait_stream(side)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            out = model(static_x, J_regressor=J)[0]
        for rep in range(3):
            g.replay()
            torch.cuda.synchronize()
            for k in eager:
                assert torch.equal(out[k], eager[k]), (rep, k)
        static_x.copy_(x2)                       # new input through the same graph
        g.replay()
        torch.cuda.synchronize()
        for k in eager2:
            assert torch.equal(out[k], eager2[k]), k
