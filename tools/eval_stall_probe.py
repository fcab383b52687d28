"""Where the evaluation product loses time outside run_clips (GPU): the phases of tepose_amd.evaluate.evaluate_clips timed with a device sync after each,
with the host-side array preparation done by torch CPU ops (the round-4 code: what stalls) -- run it with the default environment and with
OMP_NUM_THREADS=4.  Last lines: the product itself (numpy host preparation), 6 repetitions.  profiles/r05_eval_stalls.txt."""
import sys, os, time, torch, json
sys.path.insert(0, ".")
import bench
from tepose_amd import synth
from tepose_amd.testing import build_model
from tepose_amd.data import split_db_into_clips, synthetic_eval_db
from tepose_amd.evaluate import clip_metric_record
from tepose_amd.driver import run_clips
from tepose_amd.smpl import SMPL
from tepose_amd.vibe import VIBE
dev = torch.device("cuda", 0)
smpl_np = synth.synthetic_smpl(0)
L, H, T = 2, 1024, 6
model, state, _ = build_model(L, H, seed=0, device=dev, smpl_np=smpl_np, seqlen=6)
lens = (300 + 1500 * synth.uniform01('evalclips', 37)).astype(int)
db, pse = synthetic_eval_db(list(lens), seed=0)
clips = split_db_into_clips(db, pse)
vstate = synth.synthetic_vibe_state_dict(L, H, 1)
mean_v = {'pose': vstate['regressor.init_pose'][0], 'shape': vstate['regressor.init_shape'][0], 'cam': vstate['regressor.init_cam'][0]}
vibe = VIBE(seqlen=T, n_layers=L, hidden_size=H, add_linear=True, use_residual=True, pretrained='', smpl=SMPL.from_tables(smpl_np), smpl_mean_params=mean_v)
sd = vibe.state_dict()
for k, v in vstate.items():
    sd[k] = torch.from_numpy(v)
vibe.load_state_dict(sd)
vibe = vibe.to(dev).eval()
J = torch.from_numpy(smpl_np['J_regressor_h36m'])
names = list(clips.keys())


def tick(t0):
    torch.cuda.synchronize()
    return time.perf_counter() - t0


for rep in range(3):
    for cache in (True, False):
        ph = {}
        t0 = time.perf_counter()
        feats = [torch.as_tensor(clips[n]['features'], dtype=torch.float32, device=dev) for n in names]
        inits = [torch.as_tensor(clips[n]['theta_pseu'][:T - 1], dtype=torch.float32, device=dev) for n in names]
        ph['h2d'] = tick(t0); t0 = time.perf_counter()
        with torch.no_grad():
            boot = vibe(torch.stack([f[:T] for f in feats]), J_regressor=J)[-1]
        ph['vibe'] = tick(t0); t0 = time.perf_counter()
        seq = run_clips(model, feats, inits, T, J_regressor=J, keep=('kp_3d', 'verts'), cache_projections=cache)
        ph['run_clips'] = tick(t0); t0 = time.perf_counter()
        recs = []
        from tepose_amd import metrics as M
        sub = {}
        def acc(k, t0):
            torch.cuda.synchronize(); d = time.perf_counter() - t0; sub[k] = sub.get(k, 0.) + d
            if d > 0.02:
                print('      stall %.1f ms in %s of clip %d; device allocs %d' % (d * 1e3, k, s, torch.cuda.memory_stats()['num_device_alloc']))
        for s, n in enumerate(names):
            clip = clips[n]
            t1 = time.perf_counter()
            pj = torch.cat([boot['kp_3d'][s, :T - 1], seq[s]['kp_3d']], dim=0)
            pv = torch.cat([boot['verts'][s, :T - 1], seq[s]['verts']], dim=0)
            acc('cat', t1); t1 = time.perf_counter()
            target = torch.as_tensor(clip['joints3D'], dtype=torch.float32, device=dev)[:pj.shape[0]]
            target = target[:, torch.tensor(M.SPIN_TO_COMMON, device=dev)]
            acc('target', t1); t1 = time.perf_counter()
            m = M.joint_metrics(pj, target, 'lsp')
            acc('joint_metrics', t1); t1 = time.perf_counter()
            nn_ = pv.shape[0]
            tt = torch.cat([torch.zeros(nn_, 3), torch.as_tensor(clip['pose'], dtype=torch.float32)[:nn_], torch.as_tensor(clip['shape'], dtype=torch.float32)[:nn_]], dim=1).to(dev)
            acc('tt', t1); t1 = time.perf_counter()
            gv = M.gt_vertices(model, tt)
            acc('gt_vertices', t1); t1 = time.perf_counter()
            mp = M.vertex_metric(pv, gv)
            acc('vertex_metric', t1); t1 = time.perf_counter()
            recs.append(M.clip_record(s, m, mpvpe=mp))
            acc('clip_record', t1)
        print('   ', {k: '%.1f' % (v * 1e3) for k, v in sub.items()})
        out = torch.stack(recs)
        ph['metrics'] = tick(t0)
        print(rep, 'cache' if cache else 'nocache', {k: '%.1f ms' % (v * 1e3) for k, v in ph.items()}, 'reserved %.2f GB' % (torch.cuda.memory_reserved() / 2**30), 'device allocs %d frees %d' % (torch.cuda.memory_stats()['num_device_alloc'], torch.cuda.memory_stats()['num_device_free']), flush=True)
        del recs, seq, boot, feats, out, pj, pv
    if rep == 1:
        torch.cuda.empty_cache()

from tepose_amd.evaluate import evaluate_clips, gather_and_reduce
ts = []
for rep in range(7):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    recs, mine = evaluate_clips(model, vibe, clips, T, J_regressor=J, dataset='3dpw')
    res = gather_and_reduce(recs)
    torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print('product evaluate_clips + gather_and_reduce, ms per repetition (first = warm-up):', ['%.1f' % t for t in ts], 'torch threads', torch.get_num_threads(), 'cpus', len(os.sched_getaffinity(0)))
try:
    print('cgroup cpu.max:', open('/sys/fs/cgroup/cpu.max').read().strip())
except Exception as e:
    print('cgroup cpu.max unreadable', e)
