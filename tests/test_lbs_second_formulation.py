"""CPU: a second, structurally different fp64 statement of SMPL skinning, written from the SMPL paper (Loper et al. 2015,
eqs. 2-4) instead of from smplx.lbs, against the oracle's lbs().

`smplx` (where the reference's LBS arithmetic lives, requirements.txt:7) is neither vendored nor installed and its model
files are licence-gated, so oracle.lbs() CANNOT be pinned against reference code ("parity unpinned", oracle header).  This
is the strongest pin available without it: the paper's formulation shares no code path with the oracle --
  * world transforms G_k(theta, J) by explicit 4x4 recursion over the kinematic tree, joint by joint, in homogeneous
    coordinates (rotation about the joint location: translate(-J_k) . rotate . translate(J_k) composed down the chain),
  * the rest pose removed as G'_k = G_k(theta, J) . G_k(theta*, J)^-1 with a real matrix inverse (theta* = zero pose),
  * skinning as a per-vertex Python loop  v' = sum_k w_{k,v} G'_k [T_P(v); 1],
  * blend shapes as explicit per-vertex sums,
and must agree with the vectorised oracle to 1e-12."""
import numpy as np
import pytest
import torch

from oracle import tepose_ref as O
from tepose_amd import synth


def _rot(aa):
    th = np.linalg.norm(aa)
    if th < 1e-300:
        return np.eye(3)
    k = aa / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * (K @ K)


def _world_transforms(R, J, parents):
    """G_k = prod over the ancestors a of k (root first) of  [[R_a, j_a - R_a j_a... ]]: each joint rotates its subtree
    about its own rest location: M_a = T(J_a) . Rot(R_a) . T(-J_a) applied in the PARENT's posed frame, i.e.
    G_k = G_parent(k) . T(J_k - J_parent(k)) . Rot(R_k) expressed with the root at J_0."""
    G = [None] * 24
    for k in range(24):
        local = np.eye(4)
        local[:3, :3] = R[k]
        local[:3, 3] = J[k] if parents[k] < 0 else J[k] - J[parents[k]]
        G[k] = local if parents[k] < 0 else G[parents[k]] @ local
    return G


def paper_lbs(smpl, betas, R, vert_ids):
    vt, sd, pd = smpl['v_template'], smpl['shapedirs'], smpl['posedirs']
    Jr, W, parents = smpl['J_regressor'], smpl['lbs_weights'], [int(p) for p in smpl['parents']]
    V = vt.shape[0]
    # shape blend shapes, all vertices (the joint regressor needs them): T + B_S(beta)
    v_shaped = np.array([[vt[v, c] + sum(sd[v, c, l] * betas[l] for l in range(10)) for c in range(3)] for v in range(V)])
    J = Jr @ v_shaped                                              # J(beta) = regressor applied to the shaped template
    # pose blend shapes B_P(theta) = sum_n (R_n(theta) - R_n(theta*)) P_n, theta* = zero pose => R_n(theta*) = I
    pose_feature = np.concatenate([(R[k] - np.eye(3)).reshape(-1) for k in range(1, 24)])
    G = _world_transforms(R, J, parents)
    G0 = _world_transforms([np.eye(3)] * 24, J, parents)           # zero pose: pure translations to the rest joints
    Gp = [G[k] @ np.linalg.inv(G0[k]) for k in range(24)]          # G'_k = G_k(theta) G_k(theta*)^-1
    out = np.zeros((len(vert_ids), 3))
    for n, v in enumerate(vert_ids):
        tp = v_shaped[v] + np.array([sum(pose_feature[i] * pd[i, 3 * v + c] for i in range(207)) for c in range(3)])
        acc = np.zeros(4)
        for k in range(24):
            if W[v, k] != 0.0:
                acc += W[v, k] * (Gp[k] @ np.append(tp, 1.0))
        out[n] = acc[:3]
    posed = np.array([G[k][:3, 3] for k in range(24)])
    return out, posed


@pytest.mark.parametrize('seed', [0, 1, 2])
def test_paper_formulation_agrees_with_oracle_lbs(seed):
    smpl_np = synth.synthetic_smpl(seed % 2)
    smpl = {k: np.asarray(v, dtype=np.float64) for k, v in smpl_np.items()}
    rng = np.random.RandomState(100 + seed)
    betas = rng.randn(10) * (0.5 + seed)
    aa = rng.randn(24, 3) * (0.3 + 0.5 * seed)
    aa[3] = np.array([0.0, np.pi - 1e-3, 0.0])                     # a joint next to pi
    aa[7] = 0.0                                                    # and one at rest
    R = [_rot(a) for a in aa]
    vert_ids = list(range(0, 6890, 97)) + [332, 6260, 3216, 6889]
    got, posed = paper_lbs(smpl, betas, R, vert_ids)
    s = O.smpl_tensors(smpl_np, torch.float64)
    with torch.no_grad():
        v, p = O.lbs(s, torch.from_numpy(betas)[None], torch.from_numpy(np.stack(R))[None])
    assert np.abs(v[0, vert_ids].numpy() - got).max() < 1e-12
    assert np.abs(p[0].numpy() - posed).max() < 1e-12


def test_paper_formulation_with_dense_skin_weights():
    """the > 4-non-zero skin-weight table (dense skinning fallback on the GPU) through both formulations"""
    smpl_np = synth.synthetic_smpl(0, skin_nnz=7)
    smpl = {k: np.asarray(v, dtype=np.float64) for k, v in smpl_np.items()}
    rng = np.random.RandomState(5)
    betas, R = rng.randn(10), [_rot(a) for a in rng.randn(24, 3) * 0.6]
    vert_ids = list(range(5, 6890, 211))
    got, _ = paper_lbs(smpl, betas, R, vert_ids)
    s = O.smpl_tensors(smpl_np, torch.float64)
    with torch.no_grad():
        v, _ = O.lbs(s, torch.from_numpy(betas)[None], torch.from_numpy(np.stack(R))[None])
    assert np.abs(v[0, vert_ids].numpy() - got).max() < 1e-12
