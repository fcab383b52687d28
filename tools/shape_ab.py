"""A/B of kernel variants selected by environment knobs that a handle reads at creation (TEPOSE_GRU_STATE, TEPOSE_LARGE_BATCH_KERNELS, ...), interleaved in
ONE process on ONE device (cdna_hip_programming.md rule 24): each variant is its own model on the same weights; rounds alternate.
    python tools/shape_ab.py [B] [rounds] VAR=val[,VAR=val] [VAR=val ...]     (the first variant is the baseline: '' = defaults)
Per variant: layer-0 projection ms, recurrent part ms (the library's hipEvents), encoder ms (wall), max |feat - baseline feat|."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import synthetic_windows_device  # noqa: E402
from tepose_amd import synth  # noqa: E402
from tepose_amd.testing import build_model  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
variants = sys.argv[3:] or ['', 'TEPOSE_GRU_STATE=fp32', 'TEPOSE_LARGE_BATCH_KERNELS=twoacc']
dev = torch.device('cuda', 0)
smpl_np = synth.synthetic_smpl(0)
state = synth.synthetic_state_dict(2, 1024, 0)
x = synthetic_windows_device(B, 16, 3, dev)
models = []
for v in variants:
    kv = dict(p.split('=') for p in v.split(',') if p)
    old = {k: os.environ.get(k) for k in kv}
    os.environ.update(kv)
    m, _, _ = build_model(2, 1024, seed=0, device=dev, smpl_np=smpl_np, state=state)
    with torch.no_grad():
        f = m.encoder(x)          # packs (knobs are read at tepose_create, inside build_model)
    torch.cuda.synchronize()
    for k, o in old.items():
        if o is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = o
    models.append((v or 'default', m, f.clone()))
base = models[0][2]
res = {name: [] for name, _, _ in models}
with torch.no_grad():
    for r in range(rounds):
        for name, m, _ in models:
            eng = m._engine
            eng.profile_enable(True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                f = m.encoder(x)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 3 * 1e3
            k_ms, k_n, _ = eng.profile_read()
            g_ms, g_n, _ = eng.profile_read_gru()
            eng.profile_enable(False)
            res[name].append((k_ms / max(k_n, 1), g_ms / max(g_n, 1), dt))
for name, m, f in models:
    rs = res[name]
    med = lambda i: sorted(r[i] for r in rs)[len(rs) // 2]   # noqa: E731
    mn = lambda i: min(r[i] for r in rs)                      # noqa: E731
    print('%-28s projection %.3f ms (min %.3f) | recurrent %.3f (min %.3f) | encoder %.2f (min %.2f) | max|feat - base| %.2e'
          % (name, med(0), mn(0), med(1), mn(1), med(2), mn(2), float((f - base).abs().max())), flush=True)
