"""bench.py -- windows/s of the TePose per-window inference hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], SURVEY.md 8d cfg-C): synthetic [8192,16,2133] fp32
windows per GPU, model n_layers=2 / hidden=1024 (configs/config.yaml of the reference:
SEQLEN 16, the published checkpoints' GRU size), random-init weights of that architecture,
synthetic SMPL tables of the true shapes, H36M joint-regressor path (evaluate.py:109).
One step = one TePose.forward over the batch, inputs resident in HBM before the clock
starts.  N > 1: one process per GPU (torchrun), clips/windows are independent, so ranks
share nothing on the data path (weak scaling); rank 0 packs the weights and broadcasts
the packed blob over RCCL, per-rank results are gathered to rank 0.

Prints ONE JSON line (rank 0).  `roofline` is for the dominant kernel (the layer-0
input-projection GEMM, 42 % of the FLOPs, one launch per step), timed with hipEvents on
the launch stream inside the timed region.  `cpu_baseline` is the oracle's torch-CPU
restatement of the reference op sequence (nn.GRU + Linear + LBS) on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# algorithmic work per window, L=2 H=1024
GFLOP_PER_WINDOW = {6: 0.600, 16: 1.497, 32: 2.931}
# the default (split-precision) path runs the tail linears + the regressor's three FC iterations as ONE collapsed product
# (DESIGN.md 4d): 2*(2048*3072 + 1024*2048 + 3*(2*157*1024 + 1024*1024)) FLOP become 2*157*3072 per window
COLLAPSED_GFLOP_SAVING = (2.0 * (2048 * 3072 + 1024 * 2048 + 3 * (2 * 157 * 1024 + 1024 * 1024)) - 2.0 * 157 * 3072) / 1e9


def gflop_per_window(t, split=True):
    """Algorithmic FLOPs the measured path executes per window (None when BASELINE.md has no figure for this T)."""
    if t not in GFLOP_PER_WINDOW:
        return None
    return GFLOP_PER_WINDOW[t] - (COLLAPSED_GFLOP_SAVING if split else 0.0)


PEAK_F32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md, fp32-input MFMA
PEAK_F16_MFMA_TFLOPS = 2516.6         # dense fp16/bf16 MFMA (16x the fp32-input rate, ~2.5 PF spec)
SPLIT_PRODUCTS = 3                    # fp16 MFMAs per fp32-equivalent product in csrc/gemm_h3.hip


def synthetic_windows_device(B, T, seed, device):
    """Same distribution as tepose_amd.synth.synthetic_windows, generated on the GPU."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    x = torch.zeros(B, T, 2133, device=device)
    x[:, :, :2048] = torch.randn(B, T, 2048, device=device, generator=g).abs_() * 0.5
    th = torch.cat([torch.randn(B, T, 3, device=device, generator=g) * 0.05 +
                    torch.tensor([0.9, 0.0, 0.0], device=device),
                    torch.randn(B, T, 72, device=device, generator=g) * 0.2,
                    torch.randn(B, T, 10, device=device, generator=g) * 0.5], dim=-1)
    x[:, :T - 1, 2048:] = th[:, :T - 1]
    return x


def pmc_traffic(B, T, split, kernels=None):
    """(bytes, provenance): HBM-side bytes per launch of the dominant kernel.  NOT measured in this run -- PMC counters
    need their own rocprofv3 passes -- but read from the newest committed pass of this same command
    (profiles/rNN_traffic_{split,exact}.json; FETCH_SIZE x2 + WRITE_SIZE, MI355X_MICROARCH.md HBM section).
    (None, reason) when the workload differs from the profiled one -- or when the profile is STALE: it names another kernel
    symbol than the one the running library launches for this handle (`kernels` = Engine.kernel_info())."""
    import glob
    pmc_traffic.gru = None
    if (B, T) != (8192, 16):
        return None, 'no PMC pass committed for this batch / window length'
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_traffic_%s.json' % ('split' if split else 'exact'))))
    if not files:
        return None, 'no PMC pass committed'
    with open(files[-1]) as f:
        d = json.load(f)
    rel = os.path.relpath(files[-1], ROOT)
    if kernels:
        def names(entry, sym):                   # the profile entry is about this symbol (template arguments compared without blanks)
            return entry is not None and sym.replace(' ', '') in str(entry.get('kernel', '')).replace(' ', '')
        if not names(d.get('gru_step'), kernels.get('gru_step', '?')):
            d = dict(d, gru_step=None)
        if not names(d, kernels.get('projection', '?')):
            pmc_traffic.gru = d.get('gru_step')
            return None, ('STALE: %s profiles %r, the running library launches %r -- re-run tools/profile.sh and '
                          'profiles/summarize.py traffic' % (rel, str(d.get('kernel', ''))[:60], kernels.get('projection')))
    pmc_traffic.gru = d.get('gru_step')          # the fused GRU step's pass of the same run, for roofline_gru_steps
    return d.get('traffic_bytes_per_launch'), ('constant from %s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, '
                                               'FETCH x2-corrected; not collected in this run)' % os.path.relpath(files[-1], ROOT))


# every weight of the published architecture (n_layers=2, hidden=1024) touched once, fp32-equivalent bytes (the fp16 hi + lo
# planes are the same 4 bytes per element): W_ih / W_hh of the consumed directions and layers, tail linears, regressor FCs,
# blend-shape table, skin weights
# (default path: the tail linears and the regressor FCs are read as the collapsed 157 x 3072 map)
WEIGHT_BYTES = 4.0 * (3 * 3072 * 2133 + 3 * 3072 * 1024 + 3072 * 1024 + 2 * 3072 * 2048 + 2 * 3072 * 1024 +
                      157 * 3072 + 20670 * 218) + 6890 * 4 * 8.0


def small_batch_roofline(b, t, ms):
    """A forward of a few windows is neither HBM- nor MFMA-bound: it is a chain of 2T dependent recurrent steps plus a
    dozen dependent launches.  Both roofline fractions, with the bytes / FLOPs they are priced on."""
    bytes_min = WEIGHT_BYTES + b * (t * 2133 * 4.0 + 85 * 4 + 6890 * 12 + 14 * 20 + 216 * 4)
    gbps = bytes_min / (ms * 1e-3) / 1e9
    out = {'ms_per_forward': ms, 'windows_per_s': b / ms * 1e3,
           'algorithmic_bytes': bytes_min, 'achieved_GBps': gbps, 'frac_of_hbm_8TBps': gbps / 8000.0,
           'dependent_recurrent_steps': 2 * t, 'us_per_dependent_step_if_all_time_were_steps': ms * 1e3 / (2 * t)}
    if t in GFLOP_PER_WINDOW:
        tf = b * gflop_per_window(t) / (ms * 1e-3) / 1e3
        out['algorithmic_tflops'] = tf
        out['frac_of_split_mfma_peak'] = tf / (PEAK_F16_MFMA_TFLOPS / SPLIT_PRODUCTS)
    out['bound'] = 'latency (2T-step recurrence; weights read once: byte floor %.0f us, MFMA floor %.0f us)' % (
        bytes_min / 8e12 * 1e6, b * (gflop_per_window(t) or 0) * 1e9 / (PEAK_F16_MFMA_TFLOPS / SPLIT_PRODUCTS * 1e12) * 1e6)
    return out


def flat_scalars(res):
    """Top-level scalar copies of the secondary results (the driver's record keeps top-level values and reduces nested dicts
    to their key names): the strict-arithmetic run, the other kernels' roofline fractions, the evaluation product, the small
    shapes.  Only keys whose source is present."""
    def dig(*path):
        d = res
        for k in path:
            if not isinstance(d, dict) or k not in d or d[k] is None:
                return None
            d = d[k]
        return d
    src = {
        'value_exact_fp32': ('exact_fp32_mode', 'value'),
        'ms_per_step_exact_fp32': ('exact_fp32_mode', 'ms_per_step'),
        'frac_exact_fp32_of_f32_mfma_peak': ('exact_fp32_mode', 'whole_path_frac_of_f32_mfma_peak'),
        'projection_frac_exact_fp32': ('exact_fp32_mode', 'roofline', 'frac'),
        'projection_avg_ms': ('roofline', 'avg_ms'),
        'gru_steps_frac': ('roofline_gru_steps', 'frac'),
        'gru_steps_ms_per_forward': ('roofline_gru_steps', 'ms_per_forward'),
        'gru_steps_traffic_ratio': ('roofline_gru_steps', 'traffic_ratio'),
        'l1_projection_frac': ('roofline_l1_projections', 'frac'),
        'l1_projection_ms_per_forward': ('roofline_l1_projections', 'ms_per_forward'),
        'eval_frames_per_s': ('eval_driver', 'projection_cache', 'frames_per_s'),
        'eval_ms_per_lock_step': ('eval_driver', 'projection_cache', 'ms_per_lock_step'),
        'eval_seconds': ('eval_driver', 'projection_cache', 'seconds'),
        'eval_speedup_vs_reference_cpu_loop': ('eval_driver', 'speedup_vs_reference_cpu_loop'),
        'eval_model_critical_path_seconds': ('eval_driver', 'lockstep_cost_model', 'critical_path_seconds'),
        'eval_model_predicted_speedup_8gpu': ('eval_driver', 'lockstep_cost_model', 'predicted_speedup', '8'),
        'eval_model_predicted_over_measured_1gpu': ('eval_driver', 'lockstep_cost_model', 'predicted_over_measured_world1'),
        'cfgB_ms': ('other_shapes', 'cfgB_b64_T16', 'ms_per_forward'),
        'cfgA_ms': ('other_shapes', 'cfgA_b1_T16', 'ms_per_forward'),
        'cfgE_ms': ('other_shapes', 'cfgE_b1_T32_stream', 'ms_per_forward'),
        'lockstep_b37_T6_ms': ('other_shapes', 'b37_T6_3dpw_lockstep', 'ms_per_forward'),
        'cfgE_live_stream_p50_ms': ('other_shapes', 'cfgE_live_stream_T32', 'arrival_to_host_ms_p50'),
        'cpu_windows_per_s': ('cpu_baseline', 'value'),
        'cpu_cores': ('cpu_baseline', 'cores'),
    }
    out = {}
    for k, path in src.items():
        v = dig(*path)
        if isinstance(v, (int, float)) and not isinstance(v, bool):
            out[k] = v
    if 'cpu_windows_per_s' in out and out['cpu_windows_per_s'] > 0 and isinstance(res.get('value'), (int, float)):
        out['speedup_vs_cpu_baseline'] = res['value'] / out['cpu_windows_per_s']
        if 'value_exact_fp32' in out:
            out['speedup_exact_fp32_vs_cpu_baseline'] = out['value_exact_fp32'] / out['cpu_windows_per_s']
    return out


def cpu_model_string():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.lower().startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    import platform
    return platform.processor() or platform.machine()


def physical_cores():
    """Physical cores of the host (distinct (package, core) pairs of /proc/cpuinfo); None if that cannot be read."""
    try:
        seen, phys, core = set(), None, None
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('physical id'):
                    phys = line.split(':')[1].strip()
                elif line.startswith('core id'):
                    core = line.split(':')[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        seen.add((phys, core))
                    phys = core = None
        if phys is not None and core is not None:
            seen.add((phys, core))
        return len(seen) or None
    except OSError:
        return None


def cpu_baseline(state, smpl_np, L, T, budget_s=15.0, gpu_models=None, device=None):
    """Reference op sequence on the host cores (oracle, torch CPU), windows/s -- and, on the same 256-window
    sample, the largest absolute difference between that CPU result and each GPU numerics mode (the north-star
    acceptance criterion: vertices and theta within 1e-4)."""
    from oracle import tepose_ref as O
    from tepose_amd import synth
    Bc = 256
    x = synth.synthetic_windows(Bc, T, 4321)
    J = smpl_np['J_regressor_h36m']
    # pick the thread count that serves the reference path best on this host (all cores is
    # not always it: at B=256 the CPU GRU saturates well below 128 threads)
    all_cores = os.cpu_count() or torch.get_num_threads()
    phys = physical_cores() or all_cores
    best = (0.0, all_cores)
    sweep = []                                            # every (threads, windows/s) pair tried: SURVEY 8d asks for ALL
    # 8 ... the physical core count, and 1 (below).  NOT the logical count: 256 threads on a 128-core / 256-thread EPYC 9575F ran this
    # 64-window forward at 0.4 windows/s (oversubscribed MKL-DNN GRU): 160 s of a bench run for a point that can only lose
    for nt in sorted({min(n, phys) for n in (8, 16, 32, 64, phys)}):
        torch.set_num_threads(nt)
        O.tepose_fwd(state, smpl_np, x[:16], L, J_regressor=J, nn_gru=True)  # warm-up
        t0 = time.perf_counter()
        O.tepose_fwd(state, smpl_np, x[:64], L, J_regressor=J, nn_gru=True)
        r = 64 / (time.perf_counter() - t0)
        sweep.append([nt, r])
        if r > best[0]:
            best = (r, nt)
    cores = best[1]
    torch.set_num_threads(cores)
    n, t0 = 0, time.perf_counter()
    while True:
        ref = O.tepose_fwd(state, smpl_np, x, L, J_regressor=J, nn_gru=True)
        n += Bc
        el = time.perf_counter() - t0
        if el > budget_s:
            break
    torch.set_num_threads(1)                              # also the single-core rate
    t1 = time.perf_counter()
    O.tepose_fwd(state, smpl_np, x[:32], L, J_regressor=J, nn_gru=True)
    single = 32 / (time.perf_counter() - t1)
    torch.set_num_threads(cores)
    # the reference path at the small shapes of BASELINE.json too (config 1: one window; config 2: 64 windows),
    # each at the best of {1, 4, 8 threads, the thread count chosen above} -- a single window does not feed many cores
    small = {}
    for name, b, reps in (('b1_T%d' % T, 1, 12), ('b64_T%d' % T, 64, 3)):
        best_s = None
        for nt in sorted({1, min(4, all_cores), min(8, all_cores), cores}):
            torch.set_num_threads(nt)
            O.tepose_fwd(state, smpl_np, x[:b], L, J_regressor=J, nn_gru=True)
            ts = time.perf_counter()
            for _ in range(reps):
                O.tepose_fwd(state, smpl_np, x[:b], L, J_regressor=J, nn_gru=True)
            ms = (time.perf_counter() - ts) / reps * 1e3
            if best_s is None or ms < best_s[0]:
                best_s = (ms, nt)
        small[name] = {'ms_per_forward': best_s[0], 'windows_per_s': b / best_s[0] * 1e3, 'threads': best_s[1]}
    torch.set_num_threads(cores)
    res = {'value': n / el, 'unit': 'windows/s', 'cores': cores, 'kind': 'port',
           'single_thread_value': single, 'host_cpus': all_cores, 'physical_cores': phys, 'cpu_model': cpu_model_string(),
           'thread_sweep': sweep + [[1, single]],        # [threads, windows/s] on 64-window batches; `cores` = the best of them
           'small_shapes': small,
           'sample': '%d windows of [%d,2133] in batches of %d, torch %s CPU, nn.GRU op sequence, %.1f s'
                     % (n, T, Bc, torch.__version__, el)}
    if gpu_models:
        xd = torch.from_numpy(x).to(device)
        Jd = torch.from_numpy(J)
        agree = {}
        for name, mdl in gpu_models.items():
            with torch.no_grad():
                o = mdl(xd, J_regressor=Jd)[0]
            agree[name] = {k: float((o[k].cpu().double() - torch.as_tensor(ref[k]).double().reshape(o[k].shape)).abs().max())
                           for k in ('verts', 'kp_3d', 'theta')}
        res['gpu_max_abs_diff_vs_this_cpu_result'] = agree
        res['tolerance'] = 1e-4
    return res



def eval_driver_block(model, state, smpl_np, device, L, H, cpu_budget_s=8.0, with_cpu=True):
    """The real evaluation product on the driver's clock (reference flow evaluate.py:209-462): 37 synthetic
    clips of 300-1800 frames (3DPW-test-sized), seqlen 6 (evaluate.py:141), the published architecture -- VIBE bootstrap over the
    first T frames of every clip -> `run_clips` (all clips in lock-step, layer-0 projection cache) -> joint conversion, pelvis
    alignment, MPJPE / PA-MPJPE / accel / MPVPE on the device -> one record per clip -> frame-weighted means.  Timed end to end
    (host clock, synchronised), once with the projection cache (default) and once with TEPOSE_DRIVER_CACHE=0; beside it the
    reference's per-clip CPU loop (evaluate.py:247-269: one window at a time) through oracle.run_clip on a bounded prefix of
    one clip.  Outside the headline timed region; informational."""
    from tepose_amd import synth
    from tepose_amd.data import split_db_into_clips, synthetic_eval_db
    from tepose_amd.evaluate import evaluate_clips, gather_and_reduce
    from tepose_amd.smpl import SMPL
    from tepose_amd.vibe import VIBE
    T, n_clips = 6, 37
    lens = (300 + 1500 * synth.uniform01('evalclips', n_clips)).astype(int)
    db, pse = synthetic_eval_db(list(lens), seed=0)
    clips = split_db_into_clips(db, pse)
    vstate = synth.synthetic_vibe_state_dict(L, H, 1)
    mean_v = {'pose': vstate['regressor.init_pose'][0], 'shape': vstate['regressor.init_shape'][0], 'cam': vstate['regressor.init_cam'][0]}
    vibe = VIBE(seqlen=T, n_layers=L, hidden_size=H, add_linear=True, use_residual=True, pretrained='',
                smpl=SMPL.from_tables(smpl_np), smpl_mean_params=mean_v)
    sd = vibe.state_dict()
    for k, v in vstate.items():
        sd[k] = torch.from_numpy(v)
    vibe.load_state_dict(sd)
    vibe = vibe.to(device).eval()
    J = torch.from_numpy(smpl_np['J_regressor_h36m'])
    frames, steps = int(lens.sum()), int(lens.max()) - T + 1
    out = {'clips': n_clips, 'frames': frames, 'seqlen': T, 'lock_steps': steps,
           'flow': 'VIBE bootstrap -> run_clips (lock-step windows, theta feedback) -> device metrics -> per-clip records -> means'}
    old = os.environ.get('TEPOSE_DRIVER_CACHE')
    try:
        for key, env in (('projection_cache', None), ('no_cache', '0')):
            if env is None:
                os.environ.pop('TEPOSE_DRIVER_CACHE', None)
            else:
                os.environ['TEPOSE_DRIVER_CACHE'] = env
            best = None
            for rep in range(3):                      # first pass warms allocations / packs the bootstrap model
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                recs, mine = evaluate_clips(model, vibe, clips, T, J_regressor=J, dataset='3dpw')
                res = gather_and_reduce(recs)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                if rep and (best is None or dt < best):
                    best = dt
            out[key] = {'seconds': best, 'frames_per_s': frames / best, 'ms_per_lock_step': best / steps * 1e3,
                        'metrics_mm_on_random_weights': res}
            del recs
            torch.cuda.empty_cache()
    finally:
        if old is None:
            os.environ.pop('TEPOSE_DRIVER_CACHE', None)
        else:
            os.environ['TEPOSE_DRIVER_CACHE'] = old
    # the lock-step cost model behind the clip partitioner (tepose_amd.distributed.StepCost): its table re-measured on this box, its prediction for THIS
    # database at world 1 next to the measurement above, and its projection for 2 / 4 / 8 GPUs -- MODEL, not measurement (the driver measures the real curve)
    from tepose_amd.distributed import StepCost, predicted_scaling
    from tepose_amd.evaluate import measure_step_ms
    table = measure_step_ms(model, T, J_regressor=J)
    out['step_ms_table'] = {str(k): v for k, v in table.items()}
    out['lockstep_cost_model'] = predicted_scaling([int(n) for n in lens], T, StepCost(table))
    out['lockstep_cost_model']['predicted_over_measured_world1'] = out['lockstep_cost_model']['predicted_seconds']['1'] / out['projection_cache']['seconds']
    if with_cpu:
        # the reference's loop for ONE clip on the host cores: B = 1 windows, strictly serial (best thread count of 1 / 4 / 8)
        from oracle import tepose_ref as O
        name0 = next(iter(clips))
        c0 = clips[name0]
        best = None
        threads_before = torch.get_num_threads()
        for nt in (1, 4, 8):
            torch.set_num_threads(nt)
            O.run_clip(state, smpl_np, c0['features'][:T + 1], c0['theta_pseu'][:T - 1], T, L, J_regressor=smpl_np['J_regressor_h36m'])
            t0 = time.perf_counter()
            nwin = 0
            while time.perf_counter() - t0 < cpu_budget_s / 3:
                O.run_clip(state, smpl_np, c0['features'][:T + 7], c0['theta_pseu'][:T - 1], T, L, J_regressor=smpl_np['J_regressor_h36m'])
                nwin += 8
            ms = (time.perf_counter() - t0) / nwin * 1e3
            if best is None or ms < best[0]:
                best = (ms, nt)
        torch.set_num_threads(threads_before)
        windows = int(sum(max(int(n) - T + 1, 0) for n in lens))
        out['reference_cpu_loop'] = {'ms_per_window': best[0], 'threads': best[1], 'kind': 'port (oracle.run_clip: evaluate.py:247-269, one window at a time)',
                                     'sample': '8-window prefixes of one clip for %.1f s per thread count' % (cpu_budget_s / 3),
                                     'projected_seconds_for_this_database': windows * best[0] / 1e3,
                                     'frames_per_s': 1e3 / best[0]}
        out['speedup_vs_reference_cpu_loop'] = out['reference_cpu_loop']['projected_seconds_for_this_database'] / out['projection_cache']['seconds']
    return out


def self_launch(n):
    """One process per GPU through torch.distributed.run on 127.0.0.1 (the driver's own launch line), as children
    of this GPU-free process.  Returns the launcher's exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault('OMP_NUM_THREADS', '8')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def _t(msg, t0=[time.perf_counter()]):
    """phase timing on stderr (rank 0 only prints JSON on stdout)"""
    now = time.perf_counter()
    sys.stderr.write('[bench %7.1f s] %s\n' % (now - t0[0], msg))
    sys.stderr.flush()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--batch', type=int, default=8192, help='windows per GPU per step')
    ap.add_argument('--seqlen', type=int, default=16)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extra', action='store_true', help='skip the informational small-batch shapes (use under rocprofv3 --pmc)')
    ap.add_argument('--backend', default='nccl', help='torch.distributed backend (nccl = RCCL; gloo only to test the multi-process logic on a 1-GPU box)')
    ap.add_argument('--share-device0', action='store_true', help='testing only: every rank uses cuda:0')
    ap.add_argument('--force-dist', action='store_true',
                    help='run the process-group path at world size 1 too: init_process_group (nccl = RCCL), blob broadcast, '
                         'all_reduce, gather -- so that the collectives of the N-GPU run execute on a 1-GPU box')
    ap.add_argument('--rendezvous-only', action='store_true',
                    help='no GPU work: every rank joins the process group, runs the bench\'s collective sequence (barrier, all_reduce MAX, '
                         'gather of per-rank records) on host tensors and rank 0 prints a JSON line -- keeps the N-rank launch path testable '
                         'on a machine without GPUs (use with --backend gloo)')
    args = ap.parse_args()
    # the host driver of this pool only supports dmabuf IPC (RCCL and cross-process tensor sharing fail without it); the runtime
    # reads the variable when HIP initialises, so it has to be in the environment before the first torch.cuda call
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks ourselves.  This process has not touched the GPU
        # (no HIP call, no torch.cuda query) and never will: the ranks are CHILD processes of the stock launcher,
        # rank 0 prints the JSON line on the inherited stdout, and we exit with the launcher's code.
        raise SystemExit(self_launch(args.gpus))
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus:
        raise SystemExit('--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run, or unset WORLD_SIZE to let '
                         'bench.py start its own ranks)' % (args.gpus, world))
    if args.share_device0:
        local = 0
    if args.rendezvous_only:
        # the collective skeleton of the timed region, on host tensors: same calls, same order, no device
        import torch.distributed as dist
        if args.backend == 'nccl':
            raise SystemExit('--rendezvous-only runs without GPUs: use --backend gloo')
        if world > 1 or args.force_dist:
            if 'MASTER_ADDR' not in os.environ:
                import socket
                sk = socket.socket()
                sk.bind(('127.0.0.1', 0))
                os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(sk.getsockname()[1]), RANK='0', WORLD_SIZE='1', LOCAL_RANK='0')
                sk.close()
            dist.init_process_group(args.backend)
            dist.barrier()
            t_all = torch.tensor([1.0 + rank], dtype=torch.float64)
            dist.all_reduce(t_all, op=dist.ReduceOp.MAX)
            rec = torch.tensor([float(rank), 1.0 + rank, float(args.batch * args.steps), 1.0, float(local)], dtype=torch.float64)
            gathered = [torch.zeros_like(rec) for _ in range(world)] if rank == 0 else None
            dist.gather(rec, gathered, dst=0)
            dist.barrier()
            if rank == 0:
                print(json.dumps({'rendezvous_only': True, 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
                                  'max_over_ranks': float(t_all.item()), 'per_rank': [[float(v) for v in g.tolist()] for g in gathered],
                                  'dist_backend': dist.get_backend(), 'cuda_initialised': bool(torch.cuda.is_initialized())}))
            dist.destroy_process_group()
        else:
            print(json.dumps({'rendezvous_only': True, 'n_gpus': 1, 'steps': args.steps, 'warmup': args.warmup, 'per_rank': [[0.0]],
                              'cuda_initialised': bool(torch.cuda.is_initialized())}))
        return
    torch.cuda.set_device(local)
    device = torch.device('cuda', local)
    dist = None
    use_dist = world > 1 or args.force_dist
    if use_dist:
        import torch.distributed as dist
        if 'MASTER_ADDR' not in os.environ:           # --force-dist from a plain `python bench.py`: a one-rank rendezvous
            import socket
            sk = socket.socket()
            sk.bind(('127.0.0.1', 0))
            os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(sk.getsockname()[1]), RANK='0', WORLD_SIZE='1',
                              LOCAL_RANK='0')
            sk.close()
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)
        else:
            dist.init_process_group(args.backend)

    from tepose_amd import synth
    from tepose_amd.testing import build_model
    L, H, B, T = 2, 1024, args.batch, args.seqlen
    smpl_np = synth.synthetic_smpl(0)
    J = torch.from_numpy(smpl_np['J_regressor_h36m'])
    if rank == 0:
        model, state, _ = build_model(L, H, seed=0, device=device, smpl_np=smpl_np, seqlen=T)
        eng = model._engine
        with torch.no_grad():
            model(synthetic_windows_device(2, T, 1, device), J_regressor=J)     # packs the blob
        blob = eng.blob
    else:
        from tepose_amd.smpl import SMPL
        from tepose_amd.tepose import TePose
        mean = synth.synthetic_mean_params(0)
        model = TePose(seqlen=T, n_layers=L, hidden_size=H, pretrained='', smpl=SMPL.from_tables(smpl_np),
                       smpl_mean_params=mean).to(device).eval()
        eng = model._engine
        state = None
    bcast_ms = None
    n_ranks_seen = None
    if use_dist:
        # RCCL over xGMI, once: only the fp32 sections of the blob travel (276 of 770 MB: header, packed matrices, tables,
        # collapsed maps); every rank rebuilds the hi / lo planes from them (tepose_derive_planes: bit-identical planes)
        from tepose_amd.distributed import broadcast_model_weights, count_distinct_devices
        bc = broadcast_model_weights(model, src=0)
        bcast_ms, bcast_bytes = bc['ms'], bc['bytes']
        # all_gather of the device identities: N ranks must sit on N distinct GPUs for the line to be an N-GPU number
        n_ranks_seen, rank_devices = count_distinct_devices(device)
        blob = eng.blob
        if args.force_dist and world == 1:
            # one rank: exercise the receiving side too -- a second handle adopts a copy of the broadcast blob and must
            # reproduce rank 0's probe forward bit for bit (checked below as `adopted_blob_matches`)
            from tepose_amd.smpl import SMPL
            from tepose_amd.tepose import TePose
            twin = TePose(seqlen=T, n_layers=L, hidden_size=H, pretrained='', smpl=SMPL.from_tables(smpl_np),
                          smpl_mean_params=synth.synthetic_mean_params(0)).to(device).eval()
            rx = torch.zeros_like(blob)
            for off, n in eng.fp32_ranges():
                rx[off:off + n] = blob[off:off + n]
            twin._engine.adopt_blob(rx, twin, derive=True)

    # every rank runs the same 4-window probe: identical results prove the broadcast blob is the model
    with torch.no_grad():
        probe = model(synthetic_windows_device(4, T, 7, device), J_regressor=J)[0]
    probe_sum = float(probe['verts'].double().abs().sum().item())
    adopted_ok = None
    if use_dist and world == 1 and rank == 0:
        with torch.no_grad():
            adopted_ok = bool(torch.equal(twin(synthetic_windows_device(4, T, 7, device), J_regressor=J)[0]['verts'], probe['verts']))
        del twin
    x = synthetic_windows_device(B, T, 1234 + rank, device)
    torch.cuda.synchronize()
    _t('model packed, inputs resident')

    def step():
        with torch.no_grad():
            return model(x, J_regressor=J)[0]

    for _ in range(args.warmup):
        out = step()
    torch.cuda.synchronize()
    eng.profile_enable(True)
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    if use_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    _t('timed region done')
    k_ms, k_n, k_flops = eng.profile_read()
    p1_ms, p1_n, p1_flops = eng.profile_read_l1proj()
    g_ms, g_n, g_flops = eng.profile_read_gru()
    eng.profile_enable(False)
    finite = bool(torch.isfinite(out['verts']).all().item() and torch.isfinite(out['theta']).all().item())

    t_all = torch.tensor([elapsed], device=device, dtype=torch.float64)
    if use_dist:
        dist.all_reduce(t_all, op=dist.ReduceOp.MAX)
        rec = torch.tensor([float(rank), elapsed, float(B * args.steps), float(finite), probe_sum], device=device,
                           dtype=torch.float64)
        gathered = [torch.zeros_like(rec) for _ in range(world)] if rank == 0 else None
        dist.gather(rec, gathered, dst=0)     # per-rank records -> rank 0
    t_max = float(t_all.item())

    if rank == 0:
        windows = B * world * args.steps
        res = {
            'metric': '16-frame windows/sec (whole node)' if T == 16 else '%d-frame windows/sec (whole node)' % T,
            'value': windows / t_max, 'unit': 'windows/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': t_max / args.steps * 1e3, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None,
            # <= 120 characters, the strict-arithmetic pointer first (the driver's parse truncates long strings)
            'dtype': 'f32' if os.environ.get('TEPOSE_EXACT_FP32', '0') not in ('', '0') else
                     'f32 via 3 fp16 MFMAs on hi+lo halves (22 bits/operand, fp32 acc); exact-f32 MFMA run = value_exact_fp32',
            'data': 'synthetic',
            'config': {'workload': 'cfg-C synthetic [%d,%d,2133] fp32 windows per GPU, TePose n_layers=2 '
                                   'hidden=1024, random-init weights, synthetic SMPL tables, H36M-14 joint path'
                                   % (B, T),
                       'windows_per_gpu_per_step': B, 'seqlen': T, 'parallelism': 'clip-sharded dp%d' % world},
            'outputs_finite': finite,
        }
        if T in GFLOP_PER_WINDOW:
            res['whole_path_tflops'] = windows * gflop_per_window(T, os.environ.get('TEPOSE_EXACT_FP32', '0') in ('', '0')) / t_max / 1e3
            res['whole_path_frac_of_f32_mfma_peak'] = res['whole_path_tflops'] / (PEAK_F32_MFMA_TFLOPS * world)   # > 1 is possible in split mode
        split = os.environ.get('TEPOSE_EXACT_FP32', '0') in ('', '0')
        if k_n > 0:
            ach = k_flops / (k_ms / k_n * 1e-3) / 1e12
            if split:
                peak = PEAK_F16_MFMA_TFLOPS / SPLIT_PRODUCTS
                tr, tr_src = pmc_traffic(B, T, True, eng.kernel_info())
                res['roofline'] = {'bound': 'mfma', 'achieved': ach, 'peak': peak, 'unit': 'TFLOP/s', 'frac': ach / peak,
                                   'traffic': tr, 'traffic_source': tr_src,
                                   'algorithmic_bytes': float(B * T) * 2133 * 4 + 9216.0 * 2133 * 4 + float(B * T) * 9216 * 4,
                                   'kernel': '%s (single-accumulator split GEMM on scaled fp16 hi / lo planes, 256x256 tiles walked by 256 '
                                             'persistent workgroups; layer-0 input projection, M=%d N=9216 K=2133)'
                                             % (eng.kernel_info().get('projection', '?'), B * T),
                                   'launches': k_n, 'avg_ms': k_ms / k_n,
                                   # (what the chip sustains for the 32x32x16 form of this MFMA sequence with no operand traffic
                                   # at all was measured once in round 3 -- profiles/r03_projection_ablation.txt: 9.55 ms at
                                   # 1.68 GHz, 92 % busy; that constant is no longer mixed into this live line)
                                   'note': 'achieved = algorithmic fp32-equivalent FLOP/s; the kernel issues %d fp16 MFMAs '
                                           'per product (hi*hi, hi*lo, lo*hi; fp32 accumulate), so peak = dense fp16 '
                                           'MFMA %.1f / %d; executed MFMA rate = %.0f TFLOP/s'
                                           % (SPLIT_PRODUCTS, PEAK_F16_MFMA_TFLOPS, SPLIT_PRODUCTS, ach * SPLIT_PRODUCTS)}
            else:
                tr, tr_src = pmc_traffic(B, T, False, eng.kernel_info())
                res['roofline'] = {'bound': 'mfma', 'achieved': ach, 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                                   'frac': ach / PEAK_F32_MFMA_TFLOPS, 'traffic': tr, 'traffic_source': tr_src,
                                   'kernel': 'gemm_f32_kernel<false> (layer-0 input projection, M=%d N=9216 K=2133)'
                                             % (B * T),
                                   'launches': k_n, 'avg_ms': k_ms / k_n}
        if g_n > 0:
            gach = g_flops / (g_ms / g_n * 1e-3) / 1e12
            gpeak = PEAK_F16_MFMA_TFLOPS / SPLIT_PRODUCTS if split else PEAK_F32_MFMA_TFLOPS
            res['roofline_gru_steps'] = {'bound': 'mfma', 'achieved': gach, 'peak': gpeak,
                                         'unit': 'TFLOP/s', 'frac': gach / gpeak,
                                         'kernel': ('%s (scaled planes, single accumulator, cell update fused) + gru_first16_kernel' % eng.kernel_info().get('gru_step', '?')
                                                    if split else 'gru_step_kernel') +
                                                   ': the %d step launches of one forward (5T+1 consumed cell steps, '
                                                   'first-step matmuls skipped but counted)' % (2 * T + 1),
                                         'ms_per_forward': g_ms / g_n}
            gt = getattr(pmc_traffic, 'gru', None)
            if split and gt and (B, T) == (8192, 16):
                res['roofline_gru_steps'].update({
                    'traffic_per_3_direction_step': gt['traffic_bytes_per_launch'],
                    'algorithmic_bytes_per_3_direction_step': gt['algorithmic_bytes_per_launch'],
                    'traffic_ratio': gt['ratio'], 'traffic_source': 'same committed PMC passes as roofline.traffic'})
        if p1_n > 0 and p1_ms > 0:
            lach = p1_flops / (p1_ms / p1_n * 1e-3) / 1e12
            lpeak = PEAK_F16_MFMA_TFLOPS / SPLIT_PRODUCTS if split else PEAK_F32_MFMA_TFLOPS
            res['roofline_l1_projections'] = {'bound': 'mfma', 'achieved': lach, 'peak': lpeak, 'unit': 'TFLOP/s', 'frac': lach / lpeak,
                                              'ms_per_forward': p1_ms / p1_n,
                                              'kernel': 'the layer >= 1 input projections of one forward (hipEvents between two layers\' step sequences)'}
        if bcast_ms is not None:
            res['weight_broadcast_ms'] = bcast_ms
            res['weight_blob_MB'] = eng.packed_bytes / 1e6
            res['weight_broadcast_MB'] = bcast_bytes / 1e6      # the fp32 sections only; planes are re-derived on every rank
            res['per_rank'] = [[float(v) for v in g.tolist()] for g in gathered]
            sums = [r[4] for r in res['per_rank']]
            res['ranks_agree'] = bool(max(sums) - min(sums) <= 1e-6 * max(sums))
            res['dist_backend'] = dist.get_backend()
            res['n_ranks_seen'] = n_ranks_seen                    # distinct GPUs among the ranks (all_gather of device identities)
            res['rank_devices'] = rank_devices
            # lower bound of the broadcast on this node: the blob crosses at least one xGMI link (~153 GB/s per link,
            # MI355X_MICROARCH.md); at world 1 there is no link and the figure is the collective's fixed cost
            res['weight_broadcast_xgmi_floor_ms'] = bcast_bytes / 153e9 * 1e3 if world > 1 else 0.0
            if adopted_ok is not None:
                res['adopted_blob_matches'] = adopted_ok
        gpu_models = {'default': model}
        if world == 1 and split and not args.no_extra:
            # the same workload with every product on the exact-fp32 MFMA (TEPOSE_EXACT_FP32=1), for comparison
            os.environ['TEPOSE_EXACT_FP32'] = '1'
            model_x, _, _ = build_model(L, H, seed=0, device=device, smpl_np=smpl_np, seqlen=T, state=state)
            del os.environ['TEPOSE_EXACT_FP32']
            with torch.no_grad():
                ox = model_x(x, J_regressor=J)[0]
                torch.cuda.synchronize()
                model_x._engine.profile_enable(True)
                tx = time.perf_counter()
                for _ in range(args.steps):
                    ox = model_x(x, J_regressor=J)[0]
                torch.cuda.synchronize()
            tx = (time.perf_counter() - tx) / args.steps
            xk_ms, xk_n, xk_flops = model_x._engine.profile_read()
            model_x._engine.profile_enable(False)
            xach = xk_flops / (xk_ms / max(xk_n, 1) * 1e-3) / 1e12 if xk_n else None
            res['exact_fp32_mode'] = {'value': B / tx, 'ms_per_step': tx * 1e3,
                                      'roofline': None if xach is None else {
                                          'bound': 'mfma', 'achieved': xach, 'peak': PEAK_F32_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                                          'frac': xach / PEAK_F32_MFMA_TFLOPS, 'traffic': pmc_traffic(B, T, False)[0],
                                          'kernel': 'gemm_f32_kernel (exact-fp32 MFMA v_mfma_f32_32x32x2_f32; layer-0 input '
                                                    'projection, M=%d N=9216 K=2133), hipEvents on the launch stream' % (B * T),
                                          'launches': xk_n, 'avg_ms': xk_ms / max(xk_n, 1)},
                                      'whole_path_frac_of_f32_mfma_peak': B * GFLOP_PER_WINDOW.get(T, 0) / tx / 1e3 / PEAK_F32_MFMA_TFLOPS,
                                      'max_abs_diff_verts_vs_default': float((ox['verts'] - out['verts']).abs().max().item()),
                                      'max_abs_diff_kp3d_vs_default': float((ox['kp_3d'] - out['kp_3d']).abs().max().item())}
            del ox
            gpu_models['exact_fp32 (TEPOSE_EXACT_FP32=1)'] = model_x
            _t('exact-fp32 comparison done')
        if world == 1 and not args.no_extra:
            # the other BASELINE.json shapes, outside the timed region (informational, not `value`)
            extra = {}
            for name, (b, t, reps) in {'cfgB_b64_T16': (64, 16, 100), 'cfgA_b1_T16': (1, 16, 300), 'cfgE_b1_T32_stream': (1, 32, 300),
                                       'b1_T6': (1, 6, 300), 'b37_T6_3dpw_lockstep': (37, 6, 100)}.items():
                xe = synthetic_windows_device(b, t, 77, device)
                with torch.no_grad():
                    # queued back to back (status mode 'lazy': the fault word of the persistent kernels is read once, at
                    # the end of the loop -- how tepose_amd.driver runs them) ...
                    with eng.lazy_status():
                        for _ in range(5):
                            model(xe, J_regressor=J)
                        torch.cuda.synchronize()
                        te = time.perf_counter()
                        for _ in range(reps):
                            model(xe, J_regressor=J)
                        eng.check_status()              # syncs; raises if any of them gave up
                    ms = (time.perf_counter() - te) / reps * 1e3
                    # ... and as a drop-in caller sees one call: forward + the default per-call status check, which
                    # synchronises (the reference's callers .cpu() the outputs right after, evaluate.py:258-261)
                    te = time.perf_counter()
                    for _ in range(reps):
                        model(xe, J_regressor=J)
                    torch.cuda.synchronize()
                    ms_sync = (time.perf_counter() - te) / reps * 1e3
                extra[name] = small_batch_roofline(b, t, ms)
                extra[name]['ms_per_forward_with_per_call_status_sync'] = ms_sync
            # config 5 as the caller runs it (demo.py:238-252): ONE clip, window of 32 frames advancing a frame at a time
            # with theta feedback, through tepose_amd.driver.run_clips -- per-frame latency end to end
            from tepose_amd.driver import run_clips
            for t in (32, 16):
                nfr = 300 + t
                wv = synthetic_windows_device(1, nfr, 5, device)[0]
                feats, init = [wv[:, :2048].contiguous()], [wv[:t - 1, 2048:].contiguous()]
                run_clips(model, feats, init, t, keep=('theta', 'verts', 'kp_3d'))
                torch.cuda.synchronize()
                te = time.perf_counter()
                run_clips(model, feats, init, t, keep=('theta', 'verts', 'kp_3d'))
                torch.cuda.synchronize()
                extra['cfgE_stream_per_frame_ms' if t == 32 else 'stream_T16_per_frame_ms'] = (time.perf_counter() - te) / (nfr - t + 1) * 1e3
            # config 5 as a LIVE stream: frames arrive one at a time on the host (tepose_amd.stream.StreamSession: one captured
            # hipGraph replay per frame -- H2D of the feature, window shift, forward, theta feedback, D2H of the results).
            # Latency = host clock from "the frame's feature is in host memory" to "theta / kp_3d / verts are readable in
            # (pinned) host memory"; every frame waits for its own result, nothing is queued ahead.
            from tepose_amd.stream import StreamSession
            for t in (32, 16):
                nfr = 400 + t
                wv = synthetic_windows_device(1, nfr, 6, device)[0].cpu()
                for keep in (('theta', 'kp_3d', 'verts'), ('theta', 'kp_3d')):
                    ses = StreamSession(model, t, wv[:t - 1, :2048].contiguous(), wv[:t - 1, 2048:].contiguous(), J_regressor=None, keep=keep)
                    lat = []
                    for fidx in range(t - 1, nfr):
                        f = wv[fidx, :2048]
                        te = time.perf_counter()
                        ses.push(f)
                        lat.append((time.perf_counter() - te) * 1e3)
                    lat = sorted(lat[20:])
                    key = ('cfgE_live_stream_T32' if t == 32 else 'live_stream_T16') + ('' if 'verts' in keep else '_without_verts')
                    extra[key] = {'arrival_to_host_ms_p50': lat[len(lat) // 2], 'arrival_to_host_ms_p99': lat[int(len(lat) * 0.99)],
                                  'arrival_to_host_ms_min': lat[0], 'arrival_to_host_ms_max': lat[-1], 'frames': len(lat),
                                  'results_to_host': list(keep),
                                  'how': 'StreamSession.push: one hipGraph replay + one event wait per frame, pinned host rows'}
                    del ses
            # the published checkpoints' window (DATASET.SEQLEN 6, configs/repr_*.yaml) at the headline batch: 0.600 GFLOP per window
            xe = synthetic_windows_device(B, 6, 78, device)
            with torch.no_grad():
                for _ in range(2):
                    model(xe, J_regressor=J)
                torch.cuda.synchronize()
                te = time.perf_counter()
                for _ in range(5):
                    model(xe, J_regressor=J)
                torch.cuda.synchronize()
            ms6 = (time.perf_counter() - te) / 5 * 1e3
            tf6 = B * gflop_per_window(6) / (ms6 * 1e-3) / 1e3
            extra['b%d_T6_published_window' % B] = {'ms_per_forward': ms6, 'windows_per_s': B / ms6 * 1e3, 'algorithmic_tflops': tf6,
                                                     'frac_of_split_mfma_peak': tf6 / (PEAK_F16_MFMA_TFLOPS / SPLIT_PRODUCTS),
                                                     'gflop_per_window': gflop_per_window(6)}
            del xe
            extra['note'] = ('BASELINE.json configs 1 / 2 / 5 (synthetic stand-ins: random-init weights, synthetic features; the '
                             'licence-gated 3DPW data and repr_wpw_3dpw checkpoint are absent, so MPJPE vs that checkpoint is '
                             'UNMEASURED); the recurrent layers and the regressor loop run as persistent kernels '
                             '(csrc/gru_seq.hip, csrc/reg_seq.hip)')
            res['other_shapes'] = extra
            _t('other shapes done')
            res['eval_driver'] = eval_driver_block(model, state, smpl_np, device, L, H, with_cpu=not args.no_cpu_baseline)
            _t('eval driver done')
        if world == 1 and not args.no_cpu_baseline:
            res['cpu_baseline'] = cpu_baseline(state, smpl_np, L, T, gpu_models=gpu_models, device=device)
            _t('cpu baseline done')
        res.update(flat_scalars(res))
        print(json.dumps(res))
    if use_dist:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
