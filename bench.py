#!/usr/bin/env python3
"""bench.py — throughput of the OcRF render + BEV-pool + HOA hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config NAME]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path (ocrfdet_amd.hotpath.HotPath.step) over one batch of
synthetic input (seeded) that is already resident in HBM.  Rank 0 prints ONE JSON
line; `value` = BEV voxels written per second by the whole job (both pools executed per voxel
grid), with `roofline` for the dominant kernel (HIP-event timed inside the timed region) and
`cpu_baseline` (the C oracle on this box's host cores, rank 0, N=1 only).

N > 1 (`--shard`):
  camera_frames (default) — BASELINE.json north_star: the camera-frames of ONE sample over ALL ranks
      (ocrfdet_amd.sharding.CameraFramePlan: whole frames per rank while world <= n_frames, else a group of
      world / n_frames ranks per frame with the frame's cameras dealt over it); partial fused BEVs are summed by
      a reduce_scatter inside each frame group and ONE world all_gather leaves the fused grid everywhere, both
      asynchronous beside the renders and HOA-1/2; HOA-3 replicated.  Strong scaling (the sample is fixed).
      The same line carries `pipelined_camera_frames`: the same split with step k's collectives run on a
      communication stream under step k + 1's poolings and renders (outputs one step late; strong scaling, the form
      of this split that is not bound by its own exchange latency), and `samples_layout`: every rank a whole sample
      (the reference's DDP layout, no data-path collective), weak scaling; both timed right after.
  samples — only that weak-scaling layout;  frames — one (world x n_frames)-frame sequence, one all_gather;
  cameras — round 1's layout: cameras over min(world, n_cams) ranks, one dense all_reduce.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP32_PEAK_TFLOPS = 157.3        # ibid.: peak FP32 (vector) = 256 CU x 4 SIMD-32 x 32 lanes x 2 flop x 2.4 GHz
N_SIMD, CLOCK_HZ = 256 * 4, 2.4e9
VALU_CYCLES_PER_INST = 2.0      # ibid.: a wave64 VALU op issues over 2 cycles on a SIMD-32
BLEND_FLOPS_PER_PIXEL_RECORD = 20.0      # blend flops ~= 20 x sum_tiles len x 256
PMC_FILE = os.path.join('profiles', 'r6_pmc_mean.csv')     # rocprofv3 --pmc passes of this same command (tools/collect_profiles.sh)
PMC_META = os.path.join('profiles', 'r6_pmc_meta.json')    # what those passes ran: config, flags, hash of csrc/
DEFAULT_CONFIG = 'cfg2_6cam_2frame_bev200x200_render_hoa'


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--config', default=DEFAULT_CONFIG)
    ap.add_argument('--blocks', type=int, default=0,
                    help='timed blocks of --steps steps (the median block is reported); 0 = 9 blocks of up to 50 steps, else 5: a short '
                         'block is 5 ms of device time and one noisy neighbour on the host moves it by 10 %')
    ap.add_argument('--cpu-seconds', type=float, default=12.0, help='CPU-baseline sample budget')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--shard', choices=('camera_frames', 'samples', 'frames', 'cameras'), default='camera_frames')
    ap.add_argument('--backend', default='nccl', help='torch.distributed backend (nccl = RCCL); gloo only for plumbing tests')
    ap.add_argument('--no-overlap', action='store_true',
                    help='run the renders on the main stream instead of beside the pools + HOA on a side HIP stream')
    ap.add_argument('--no-graph', action='store_true',
                    help="--scope neck: issue the step kernel by kernel instead of replaying it as one hipGraph")
    ap.add_argument('--scope', choices=('hotpath', 'neck'), default='hotpath',
                    help="'hotpath': pools (+ one rendered view per camera + HOA) — the headline step; 'neck': the whole "
                         "OcRFViewTransformerFull.view_transform (one rendered view per sample, like the reference)")
    ap.add_argument('--host-geometry', action='store_true',
                    help='--scope neck: the per-forward calibration algebra on the HOST (the reference\'s own torch calls: rank '
                         'vectors bit-exact) instead of the module\'s default, on the device when the calibration tensors are there')
    ap.add_argument('--device-geometry', action='store_true',
                    help="--scope neck --index-prep per_step: per-forward calibration algebra on the GPU (ocrf_geometry_blocks): "
                         "no device read-back, no synchronisation in the forward")
    ap.add_argument('--host-calibration', action='store_true',
                    help="--scope neck --index-prep per_step: hand the calibration tensors over as HOST tensors (the "
                         "dataloader's copies) — the forward then has no device -> host read-back at all")
    ap.add_argument('--no-pool-backends', action='store_true',
                    help='skip pools.backends_alone_us (the three pooling kernels alone: extra launches a kernel trace should not hold)')
    ap.add_argument('--no-per-step', action='store_true', help='skip the second timed loop with the index preparation inside the step')
    ap.add_argument('--dense-exchange', action='store_true',
                    help="camera_frames: step 1 as the dense reduce_scatter instead of the wedge-sparse isend / irecv round")
    ap.add_argument('--render-mode', choices=('planned', 'per_call'), default='planned',
                    help="'planned': the render's calibration-only front end (near-plane / frustum cull, depth order, projected "
                         "centres) cached per frame like the rank vectors (static render plan); 'per_call': recomputed every render")
    ap.add_argument('--render-guard', choices=('host', 'device'), default='host',
                    help="planned renders: 'host' = the plan's extent bound is verified by one status read after the timed region; "
                         "'device' = the per-call pipeline is armed behind every planned render on the GPU")
    ap.add_argument('--lss-pool', choices=('auto', 'tile', 'mfma', 'panel'), default='auto',
                    help="kernel of the LSS pooling with cached ranks: the VALU tile kernel, the MFMA panel kernel (DESIGN.md "
                         "4.1b) or the panel latency kernel (4.1c); auto = HotPath's choice (tile beside the persistent "
                         "blend, panel where nothing VALU-bound runs beside the poolings)")
    ap.add_argument('--blend-workgroups', default='auto',
                    help="planned renders beside the main stream: workgroups of the persistent blend that stay while the main "
                         "chain runs ('auto' = 3.5 per CU; DESIGN.md section 5)")
    ap.add_argument('--gaussians', choices=('init', 'stress', 'objects'), default='init',
                    help='synthetic Gaussian parameter set of the headline step (synthetic.grid_gaussians): init = the '
                         'reference heads at seeded init (every pixel saturates early: the best case), stress = SURVEY 8d\'s '
                         'second set, objects = object-centric opacity with new parameters every step.  Whatever the headline '
                         'uses, the default line carries all three under gaussian_sets')
    ap.add_argument('--no-gaussian-sets', action='store_true', help='skip the gaussian_sets leg of the default line')
    ap.add_argument('--index-prep', choices=('cached', 'per_step'), default='cached',
                    help="'per_step': rank vectors recomputed by the HIP index preparation inside every step "
                         "(the reference with accelerate=False); 'cached': once per calibration (accelerate=True)")
    return ap.parse_args()


def _hoa_params(hp):
    p = {}
    for prefix, m in hp.hoa_mods.items():
        for k, v in m.state_dict().items():
            p[f'{prefix}.{k}'] = v.detach().cpu().numpy()
    return p


def cpu_baseline(hp, depth, feat, budget_s):
    """Times the C/OpenMP oracle (and, for HOA, the numpy oracle) on a bounded sample of the same step: the two pools
    of the whole step (a few repetitions), a few rendered views of frame 0 (the reference-structured rasteriser sorts
    all tile instances, so a full step of views would take minutes) and HOA-1/2/3 of ONE frame; the step time is
    assembled as t_pools + views_per_step * t_view + frames * t_hoa.  The same sample on ONE thread (the scalar port)
    is timed in a child process with a time limit."""
    import numpy as np
    import oracle
    oracle.build()
    d, f = depth.cpu().numpy(), feat.cpu().numpy()
    plans = []
    for p in (hp.lss, hp.ht):
        plans.append(tuple(t.cpu().numpy() for t in (p.ranks_depth, p.ranks_feat, p.ranks_bev, p.starts, p.lengths)) + (p.bev_shape,))

    def pools():
        for rd, rf, rb, st, ln, shape in plans:
            oracle.bev_pool_v2(d, f, rd, rf, rb, shape, st, ln)     # includes the wrapper's permute
    # Thread count: the comparator gets the BEST of what this process may use — every core of its affinity set, and the
    # 16-core share a one-GPU box of this pool really hands out (on such a box the node shows 128+ cores but more threads
    # than the share only oversubscribe it: 128 threads rendered a view slower than one).  OCRF_CPU_THREADS pins it.
    max_threads = oracle.num_threads()
    affinity = len(os.sched_getaffinity(0))
    pinned = int(os.environ.get('OCRF_CPU_THREADS', 0))
    cap = min(max_threads, affinity)
    # the best thread count PER LEG (the rasteriser leg is 84 % of the CPU step and was never swept)
    candidates = [pinned] if pinned else sorted({cap, min(cap, 16), min(cap, 32), min(cap, 64)})
    tried = {}
    for c in candidates:
        oracle.set_num_threads(c)
        pools()                                                      # warm-up
        t0 = time.perf_counter()
        reps = 0
        while reps < 3 or time.perf_counter() - t0 < 0.5:
            pools()
            reps += 1
        tried[c] = (time.perf_counter() - t0) / reps
    pool_threads = min(tried, key=tried.get)
    oracle.set_num_threads(pool_threads)
    n, t0 = 0, time.perf_counter()
    while True:
        pools()
        n += 1
        el = time.perf_counter() - t0
        if el >= min(budget_s, 4.0) or n >= 200:
            break
    t_pools = el / n
    t_view, n_views_timed, rendered, view_args, view_threads, tried_view = 0.0, 0, None, None, None, {}
    if hp.cfg.render:
        g = hp.gauss
        Himg, Wimg = hp.cfg.input_size
        args = (hp.voxel_xyz[0].reshape(-1, 3).cpu().numpy(), g['rgb'].cpu().numpy(), g['opacity'].cpu().numpy(),
                g['scales'].cpu().numpy(), g['rotations'].cpu().numpy())
        cams = hp.render_cams
        view_args = dict(xyz=args[0], rgb=args[1], opacity=args[2], scales=args[3], rotations=args[4],
                         vm=cams['vm'][0].cpu().numpy(), pm=cams['pm'][0].cpu().numpy(), tfx=np.float64(cams['tfx'][0]),
                         tfy=np.float64(cams['tfy'][0]), H=np.int64(Himg), W=np.int64(Wimg))

        def one_view(v):
            return oracle.rasterize_forward(*args, cams['vm'][v].cpu().numpy(), cams['pm'][v].cpu().numpy(),
                                            cams['tfx'][v], cams['tfy'][v], Himg, Wimg, np.zeros(3, np.float32))
        # the rasteriser leg's own sweep: one view per candidate (a view is ~ 0.2 s), the fastest renders the sample
        tried_view = {}
        for c in candidates:
            oracle.set_num_threads(c)
            t0 = time.perf_counter()
            one_view(0)
            tried_view[c] = time.perf_counter() - t0
            if tried_view[c] > 0.25 * budget_s:          # (a count that oversubscribes the box's share: stop sweeping upwards)
                break
        view_threads = min(tried_view, key=tried_view.get)
        oracle.set_num_threads(view_threads)
        t0 = time.perf_counter()
        while True:
            v = n_views_timed % len(hp.cams)
            r = one_view(v)
            rendered = r['num_rendered']
            n_views_timed += 1
            el = time.perf_counter() - t0
            if el >= 0.5 * budget_s or n_views_timed >= 12:
                break
        t_view = el / n_views_timed
    t_hoa, hoa_note = 0.0, 'no HOA in this configuration'
    if hp.cfg.hoa:
        from oracle import hoa as ohoa
        X, Y, _ = hp.cfg.bev_xyz
        p = _hoa_params(hp)
        opac = hp.frame_gauss[0]['opacity'].cpu().numpy().astype(np.float32)
        alpha = hp.alpha_lidar[:1].cpu().numpy()
        pos = hp.bev_pos1[:1].cpu().numpy()
        geom = np.zeros((1, hp.cfg.channels, Y, X), np.float32)
        t0 = time.perf_counter()
        oa, _ = ohoa.hoa1(opac, alpha, p, hp.cfg.num_height, Y, X)
        ob = ohoa.opacity_voxel_to_bev(oa, pos, p)
        m = ohoa.opacity_mask(geom, ob, p)
        _ = geom * m
        t_hoa = time.perf_counter() - t0
        hoa_note = 'HOA-1/2/3 of one frame by the numpy oracle (oracle/hoa.py), once'
    frames = hp.batch
    t_step = t_pools + hp.views_per_step * t_view + frames * t_hoa
    single = single_thread_sample(plans, d, f, view_args, limit_s=max(20.0, 2.0 * budget_s))
    if single.get('ms_pools') is not None and single.get('ms_per_view') is not None:
        single['ms_per_step'] = single['ms_pools'] + hp.views_per_step * single['ms_per_view'] + 1e3 * frames * t_hoa
        single['value'] = hp.bev_voxels_per_step / (single['ms_per_step'] * 1e-3)
    phys = physical_cores()
    return dict(value=hp.bev_voxels_per_step / t_step, unit='BEV voxels/s',
                cores=max(pool_threads, view_threads if hp.cfg.render else 0),
                threads={'pooling_leg': pool_threads, 'rasteriser_leg': view_threads if hp.cfg.render else None,
                         'hoa_leg': 'numpy / BLAS default'},
                threads_tried={str(k): 1e3 * v for k, v in tried.items()},
                threads_tried_rasteriser={str(k): 1e3 * v for k, v in (tried_view.items() if hp.cfg.render else ())},
                box={'physical_cores': phys, 'logical_cpus': os.cpu_count(), 'affinity_set': affinity},
                threads_note='OpenMP thread counts swept PER LEG (ms per pair of pools / ms per rendered view); each leg runs '
                             'on its own fastest count; `cores` is the larger of the two',
                cores_note='OpenMP threads of the pooling, of the per-Gaussian / per-tile loops and of the per-tile '
                           'sorts of the tile instances (bucketed by tile first; the histogram pass is one thread); the '
                           'numpy HOA leg uses whatever BLAS threads numpy has',
                kind='port', ms_per_step=1e3 * t_step, ms_pools=1e3 * t_pools, ms_per_view=1e3 * t_view,
                ms_hoa_per_frame=1e3 * t_hoa,
                views_per_sec=(hp.views_per_step / t_step) if hp.views_per_step else 0.0,
                single_thread=single,
                sample=f'{n} x (LSS pool + HT pool of the whole step, same inputs and ranks as the GPU)'
                       + (f', {n_views_timed} rendered view(s) of frame 0 ({rendered} tile instances in the last)'
                          if hp.cfg.render else '') + f', {hoa_note}; step time assembled as pools + '
                       f'{hp.views_per_step} x view + {frames} x HOA; C/OpenMP + numpy oracles')


def physical_cores():
    """Distinct (package, core) pairs of /proc/cpuinfo — the box's physical core count (None if it cannot be read)."""
    try:
        cores, pkg, core = set(), None, None
        for line in open('/proc/cpuinfo'):
            if line.startswith('physical id'):
                pkg = line.split(':')[1].strip()
            elif line.startswith('core id'):
                core = line.split(':')[1].strip()
            elif not line.strip():
                if pkg is not None and core is not None:
                    cores.add((pkg, core))
                pkg = core = None
        return len(cores) or None
    except OSError:
        return None


def single_thread_sample(plans, d, f, view_args, limit_s):
    """The scalar port: the pools once and ONE rendered view on a single thread, in a child process that is stopped
    after ``limit_s`` (a 520 k-Gaussian view takes the single-threaded reference-structured rasteriser tens of seconds)."""
    import pickle
    import subprocess
    import tempfile
    import numpy as np
    out = dict(cores=1, ms_pools=None, ms_per_view=None, limit_s=limit_s)
    with tempfile.TemporaryDirectory(dir='/tmp') as tmp:
        blob = os.path.join(tmp, 'sample.pkl')
        with open(blob, 'wb') as fo:
            pickle.dump(dict(plans=plans, d=d, f=f, view=view_args), fo, protocol=4)
        code = (
            'import pickle, sys, time, json\n'
            f'sys.path.insert(0, {ROOT!r})\n'
            'import numpy as np, oracle\n'
            'oracle.set_num_threads(1)\n'
            f's = pickle.load(open({blob!r}, "rb"))\n'
            't0 = time.perf_counter()\n'
            'for rd, rf, rb, st, ln, shape in s["plans"]:\n'
            '    oracle.bev_pool_v2(s["d"], s["f"], rd, rf, rb, shape, st, ln)\n'
            'print(json.dumps({"ms_pools": 1e3 * (time.perf_counter() - t0)}), flush=True)\n'
            'v = s["view"]\n'
            'if v is not None:\n'
            '    t0 = time.perf_counter()\n'
            '    oracle.rasterize_forward(v["xyz"], v["rgb"], v["opacity"], v["scales"], v["rotations"], v["vm"], v["pm"],\n'
            '                             float(v["tfx"]), float(v["tfy"]), int(v["H"]), int(v["W"]), np.zeros(3, np.float32))\n'
            '    print(json.dumps({"ms_per_view": 1e3 * (time.perf_counter() - t0)}), flush=True)\n')
        env = dict(os.environ, OMP_NUM_THREADS='1')
        proc = subprocess.Popen([sys.executable, '-c', code], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env, text=True)
        try:
            txt, _ = proc.communicate(timeout=limit_s)
        except subprocess.TimeoutExpired:
            proc.kill()
            txt, _ = proc.communicate()
            out['note'] = f'stopped after {limit_s:.0f} s'
        for line in (txt or '').splitlines():
            try:
                out.update(json.loads(line))
            except Exception:       # noqa: BLE001
                pass
        if view_args is not None and out['ms_per_view'] is None:
            out['ms_per_view_lower_bound'] = 1e3 * limit_s - (out['ms_pools'] or 0.0)
    return out


def bench_neck(args, cfg, dev, world, rank):
    """--scope neck: one step = pre-filter + the whole view_transform of B = n_frames samples per rank
    (whole samples per rank, no data-path collective).  Roofline: the LSS/HT pooling kernel."""
    import torch.distributed as dist
    from ocrfdet_amd import _lib, hotpath
    neck = hotpath.NeckPath(cfg, dev, accelerate=args.index_prep == 'cached', seed=rank,
                            host_calibration=args.host_calibration)
    # the module's default (round 6): calibration algebra on the device whenever the calibration tensors are there;
    # --host-geometry opts out (rank vectors bit-exact with the reference's host formulation)
    device_geometry = not args.host_geometry and not args.host_calibration
    neck.module.device_geometry = None if device_geometry else False
    args.device_geometry = device_geometry
    if args.index_prep == 'cached':
        neck.module.render_guard = args.render_guard      # 'host': the armed per-call chain is not issued; checked below
    # one hipGraph replay per step: cached geometry, or the per-forward geometry inside the graph (device geometry)
    graphed = (args.index_prep == 'cached' or device_geometry) and not args.no_graph
    for _ in range(args.warmup):
        neck.step()
    step = neck.step
    if graphed:
        # the whole step as ONE hipGraph launch; the random camera of each sample is staged into static
        # device tensors before every replay (hotpath.NeckPath.capture)
        neck.capture()
        step = neck.step_graphed
        for _ in range(3):
            step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier(group=CONTROL)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier(group=CONTROL)
    elapsed = time.perf_counter() - t0
    neck.module.check_render()        # render_guard='host': one status read verifies every render of the timed region
    # kernel events cannot bracket launches inside a graph replay: the pooling kernel is timed over a
    # short eager run of the same step right after the timed region
    timer = _lib.KernelTimer(_lib.K_BEV_POOL_FWD, 64)
    torch.cuda.synchronize()
    timer.arm()
    for _ in range(min(16, args.steps)):
        neck.step()
    torch.cuda.synchronize()
    timer.disarm()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=CONTROL)
        elapsed = float(t.item())
    if rank == 0:
        ms = timer.read_ms()
        avg_ms = sum(ms) / max(len(ms), 1)
        m, geo = neck.module, neck.module._geo
        d_numel = neck.batch * cfg.n_cams * cfg.D * cfg.feat_hw[0] * cfg.feat_hw[1]
        f_numel = neck.batch * cfg.n_cams * cfg.channels * cfg.feat_hw[0] * cfg.feat_hw[1]
        alg = None
        if geo is not None and len(geo.lss) == 5 and geo.lss[0] is not None:
            X, Y, Z = cfg.bev_xyz
            per = [4 * (d_numel + f_numel + 3 * r[0].numel() + 2 * r[3].numel() + neck.batch * z * Y * X * cfg.channels)
                   for r, z in ((geo.lss, Z), (geo.ht, 1))]
            alg = 0.5 * sum(per)
        achieved = alg / (avg_ms * 1e-3) / 1e9 if alg and avg_ms > 0 else None
        out = {'metric': 'BEV voxels/sec + rendered views/sec, 6-cam 256x704',
               'value': neck.bev_voxels_per_step * world * args.steps / elapsed, 'unit': 'BEV voxels/s',
               'rendered_views_per_sec': neck.views_per_step * world * args.steps / elapsed,
               'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * elapsed / args.steps,
               'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
               'config': {'workload': cfg.name + ' / whole neck', 'cams': cfg.n_cams, 'frames_per_gpu': cfg.n_frames,
                          'bev': list(cfg.bev_xyz), 'channels': cfg.channels, 'depth_bins': cfg.D,
                          'stages': 'prefilter+lss_pool+ht_pool+colour/alpha sampling+gauss heads+nerf branch+render+hoa+bev fusion',
                          'views_per_step': neck.views_per_step, 'render_camera': 'reference',
                          'launch': 'one hipGraph replay per step' if graphed else 'eager (kernel by kernel)',
                          'index_prep': 'cached (accelerate=True)' if args.index_prep == 'cached' else
                          'per step, HIP (accelerate=False); calibration ' +
                          ('algebra on the device (ocrf_geometry_blocks), no read-back' if args.device_geometry else
                           'handed over as host tensors' if args.host_calibration else 'read back from the device (one packed copy)'),
                          'sharding': 'none' if world == 1 else f'{world} ranks x whole samples, no data-path collective'},
               'roofline': {'bound': 'hbm', 'kernel': timer.kernel_name, 'achieved': achieved, 'peak': HBM_PEAK_GBS,
                            'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS if achieved else None, 'traffic': None,
                            'algorithmic_bytes_per_launch': alg, 'avg_launch_us': 1e3 * avg_ms, 'launches_timed': len(ms),
                            'note': 'kernel timed over an eager run of the same step right after the timed region '
                                    '(HIP events cannot bracket a launch inside a graph replay)'},
               'cpu_baseline': None}
        emit(out)
    timer.close()
    if world > 1:
        try:
            dist.destroy_process_group()
        except Exception:       # noqa: BLE001
            pass


def gaussian_set_figures(cfg, dev, kind, depth, feat, steps=20, blocks=5, hp_kw=None, plan_bins='default'):
    """The planned step on Gaussian set ``kind`` with NEW parameters every step (two parameter sets per frame taken in
    turn: the blend's adaptive head always works from what the OTHER set needed) -> dict: ms_per_step (median block),
    blend_us (HIP events on the blend's stream inside those steps), and from the instrumented build of the same kernel
    (one launch after the timed region, parameters of phase 0 behind a phase-1 step): list entries scanned / records staged /
    wave-records evaluated per tile pair, p50 / p99 / max of the entries a tile pair scanned (its reach into the view's
    list), the per-view heads the blend left behind, executed pixel.records."""
    import numpy as np
    from ocrfdet_amd import _lib, hotpath
    hp = hotpath.HotPath(cfg, dev, gaussians=kind, alternate=True, **(hp_kw or {}))
    if plan_bins != 'default':
        hp.plan_bins = plan_bins
    k = [0]

    def step():
        hp.set_phase(k[0] & 1)
        k[0] += 1
        return hp.step(depth, feat)
    for _ in range(8):
        step()
    torch.cuda.synchronize()
    tb = _lib.KernelTimer(_lib.K_RASTER_BLEND_SORTED, 2 * steps + 8)
    tb.arm()
    ts = [timed(step, steps, 1, dev)]
    _lib.KernelTimer.disarm_all()
    blend_ms = tb.mean_ms()
    tb.close()
    for _ in range(blocks - 1):
        ts.append(timed(step, steps, 1, dev))
    hp.check_render_plans()
    out = {'ms_per_step': 1e3 * sorted(ts)[len(ts) // 2] / steps, 'blend_us': 1e3 * blend_ms if blend_ms else None,
           'one_call': bool(hp._compiled), 'parameters': 'two sets per frame, alternating every step'}
    # the same kernel alone (no other stream beside it)
    was = hp.overlap
    hp.overlap = False
    ib = _lib.KernelTimer(_lib.K_RASTER_BLEND_SORTED, 64)
    ib.arm()
    for _ in range(12):
        step()
    torch.cuda.synchronize()
    _lib.KernelTimer.disarm_all()
    out['blend_alone_us'] = 1e3 * ib.mean_ms() if ib.mean_ms() else None
    ib.close()
    hp.overlap = was
    H, W = cfg.input_size
    tiles = ((W + 15) // 16) * ((((H + 15) // 16) + 1) // 2)
    scanned, staged, evald, heads, phases = [], [], 0.0, [], []
    hp.set_phase(1)
    hp.step(depth, feat)
    hp.set_phase(0)
    for entry in hp._plans():
        plan = entry[0]
        n_wg = tiles * entry[2] * len(hp.cams)
        buf = torch.zeros(n_wg * 4 * 8, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        _lib.lib().ocrf_diag_plan_stats(_lib.ptr(buf))
        try:
            hp._render_planned(entry)
            torch.cuda.synchronize()
        finally:
            _lib.lib().ocrf_diag_plan_stats(None)
        st = buf.view(n_wg, 4, 8).cpu().numpy().astype(np.float64)
        scanned.append(st[:, 0, 3])
        staged.append(st[:, 0, 4])
        phases.append(st[:, :, :3].reshape(-1, 3))
        evald += float(st[:, :, 6].sum())
        ctl = plan._dyn[plan._dyn.numel() - 1024:].view(torch.int32).cpu().numpy()      # control words (raster_plan.hip kCtl*)
        heads += [int(v) for v in ctl[32:32 + plan.V]]
    hp.check_render_plans()
    sc, sg = np.concatenate(scanned), np.concatenate(staged)
    n_launch = len(hp._plans())
    out.update({'scanned_per_tile_pair': float(sc.mean()), 'staged_per_tile_pair': float(sg.mean()),
                'evaluated_wave_records_per_tile_pair': evald / len(sc),
                'reach_p50': float(np.percentile(sc, 50)), 'reach_p99': float(np.percentile(sc, 99)), 'reach_max': float(sc.max()),
                'head': heads, 'list_entries_per_view': [int(v) for e in hp._plans() for v in (e[0].kept or [])],
                'executed_pixel_records_per_launch': evald * 128.0 / n_launch})
    ph = np.concatenate(phases)
    out['tile_pair_us'] = float(ph.sum(1).mean()) / 100.0            # (s_memrealtime: 100 MHz)
    out['phase_shares_scan_stage_blend'] = [round(float(x), 3) for x in ph.mean(0) / ph.mean(0).sum()]
    if out['blend_alone_us']:
        out['blend_alone_ns_per_kilo_pixel_record'] = 1e6 * out['blend_alone_us'] / out['executed_pixel_records_per_launch']
    del hp
    return out


def source_hash():
    """sha256 over the library's sources (csrc/*.hip, *.h): the counters of a profile belong to ONE build."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, 'ocrfdet_amd', 'csrc', '*.hip')) +
                    glob.glob(os.path.join(ROOT, 'ocrfdet_amd', 'csrc', '*.h'))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, 'rb').read())
    return h.hexdigest()[:16]


_PMC_OK = None


def pmc_valid(args):
    """The committed counters are used only for the run they were collected on: same config, same render mode and
    the same library sources (profiles/r5_pmc_meta.json); anything else reports no traffic / counters."""
    global _PMC_OK
    if _PMC_OK is None:
        _PMC_OK = False
        try:
            meta = json.load(open(os.path.join(ROOT, PMC_META)))
            _PMC_OK = (meta.get('config') == args.config and meta.get('render_mode') == args.render_mode and
                       meta.get('source_hash') == source_hash())
        except Exception:       # noqa: BLE001
            _PMC_OK = False
    return _PMC_OK


def pmc_counters(kernel_prefix, args=None):
    """Mean per launch of every counter rocprofv3 collected for one kernel (profiles/r5_pmc_mean.csv, written by
    tools/collect_profiles.sh from separate --pmc passes of this command with --no-overlap): {counter: mean}."""
    path = os.path.join(ROOT, PMC_FILE)
    out = {}
    if not os.path.exists(path) or (args is not None and not pmc_valid(args)):
        return out
    import csv
    for r in csv.DictReader(open(path)):
        if r['kernel'].startswith(kernel_prefix) and int(r['launches']) > 4:     # skip the one-off variants
            out[r['counter']] = float(r['mean'])
    return out


def hbm_traffic(c):
    """HBM bytes per launch from the PMC passes: FETCH_SIZE / WRITE_SIZE are KiB; FETCH_SIZE counts half of a wide
    coalesced read stream on gfx950 (MI355X_MICROARCH.md, HBM) — doubled."""
    if 'FETCH_SIZE' not in c or 'WRITE_SIZE' not in c:
        return None
    return (2.0 * c['FETCH_SIZE'] + c['WRITE_SIZE']) * 1024.0


CONTROL = None      # gloo group of all ranks: barriers, the max-over-ranks of the clock, agreement on errors


def agree(ok, world):
    """True iff every rank says ok (control plane, CPU): a rank that caught an exception must not leave the
    others inside a collective."""
    if world == 1:
        return bool(ok)
    import torch.distributed as dist
    t = torch.tensor([1 if ok else 0], dtype=torch.int32)
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=CONTROL)
    return bool(t.item())


def _poll(done, seconds):
    """True once done() says so, False after `seconds` — never blocks in a collective."""
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        if done():
            return True
        time.sleep(0.01)
    return False


def rccl_probe(world, rank, dev, groups, seconds=30.0):
    """Tiny collectives of every kind the camera-frame exchange uses, on the world and on this rank's frame group, each
    waited for by POLLING with a time limit: -> (error text or None, hung).  An exception leaves RCCL usable for nobody
    but harms nothing (every rank then agrees, over gloo, to run the samples layout, which has no data-path
    collective); a collective that never completes leaves a kernel spinning on the device — the caller exits."""
    import torch.distributed as dist
    try:
        t = torch.ones(256, device=dev)
        w = dist.all_reduce(t, async_op=True)
        if not _poll(w.is_completed, seconds):
            return 'world all_reduce did not complete', True
        if abs(float(t[0].item()) - world) > 1e-3:
            return 'world all_reduce returned a wrong sum', False
        out = torch.empty(256 * world, device=dev)
        w = dist.all_gather_into_tensor(out, t, async_op=True)
        if not _poll(w.is_completed, seconds):
            return 'world all_gather did not complete', True
        for ranks, group in groups.items():
            if rank not in ranks:
                continue
            G = len(ranks)
            src, dst = torch.ones(G * 64, device=dev), torch.empty(64, device=dev)
            w = dist.reduce_scatter_tensor(dst, src, group=group, async_op=True)
            if not _poll(w.is_completed, seconds):
                return f'reduce_scatter in group {list(ranks)} did not complete', True
            if abs(float(dst[0].item()) - G) > 1e-3:
                return f'reduce_scatter in group {list(ranks)} returned a wrong sum', False
            ops, bufs = [], []
            for r in ranks:
                if r != rank:
                    bufs.append(torch.empty(64, device=dev))
                    ops.append(dist.P2POp(dist.isend, src[:64], r, group))
                    ops.append(dist.P2POp(dist.irecv, bufs[-1], r, group))
            for q in (dist.batch_isend_irecv(ops) if ops else []):
                if not _poll(q.is_completed, seconds):
                    return f'isend / irecv in group {list(ranks)} did not complete', True
        torch.cuda.synchronize()
        return None, False
    except Exception as e:       # noqa: BLE001
        return f'{type(e).__name__}: {e}'[:300], False


def timed(step, steps, world, dev):
    """steps x step() bracketed by a barrier + torch.cuda.synchronize() on both sides; the MAX over ranks."""
    import torch.distributed as dist
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier(group=CONTROL)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier(group=CONTROL)
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=CONTROL)
        elapsed = float(t.item())
    return elapsed


def timed_median(step, steps, world, dev, blocks=3):
    """Median of `blocks` timed blocks: the side measurements (per-step / per-sample modes) are single figures, and one
    block now and then catches a 50-100 ms hiccup of the box (seen twice at configs[1]: 1.6 ms instead of 0.18)."""
    return sorted(timed(step, steps, world, dev) for _ in range(blocks))[blocks // 2]


_REAL_STDOUT = None


def emit(obj):
    """The ONE JSON line of the contract, on the process's real stdout."""
    line = (json.dumps(obj) + '\n').encode()
    if _REAL_STDOUT is None:
        sys.stdout.write(line.decode())
        sys.stdout.flush()
    else:
        os.write(_REAL_STDOUT, line)


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    # Everything but the result line goes to stderr: gloo (the control group of N > 1 runs) reports its connections on
    # the C-level stdout, torch.distributed.run forwards every rank's stdout — the contract is ONE line.
    global _REAL_STDOUT
    sys.stdout.flush()
    _REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    single_dev = os.environ.get('OCRF_BENCH_SINGLE_DEVICE') == '1'      # plumbing test: all ranks on cuda:0 (gloo)
    if single_dev:
        local_rank = 0
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        import datetime
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank),
                                    timeout=datetime.timedelta(seconds=120))
        else:
            dist.init_process_group(args.backend)
        global CONTROL
        CONTROL = dist.new_group(backend='gloo', timeout=datetime.timedelta(seconds=120))
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)

    from ocrfdet_amd import _lib, bevpool, hotpath, sharding, synthetic
    _lib.lib()                      # raises if libocrf_hip.so is missing — no fallback
    cfg = synthetic.CONFIGS[args.config]
    shard = args.shard if world > 1 else 'none'
    if args.scope == 'neck':
        return bench_neck(args, cfg, dev, world, rank)
    X, Y, Z = cfg.bev_xyz
    rkw = dict(render_mode=args.render_mode, render_guard=args.render_guard)
    planned = cfg.render and args.render_mode == 'planned'
    k_blend = _lib.K_RASTER_BLEND_SORTED if planned else _lib.K_RASTER_BLEND
    active = True
    sp = None
    shard_error = None
    if shard == 'camera_frames':
        # the exchange has only ever run over gloo on the build box (one GPU): if RCCL refuses any part of it, every
        # rank falls back to the samples layout TOGETHER and the line says so, instead of one rank dying in a collective
        hung = False
        use_rccl = args.backend == 'nccl'
        try:
            if use_rccl:
                # tiny collectives of every kind the exchange uses, polled with a time limit, BEFORE the first real one:
                # the world first (the exchange's constructor tries its in-place gather on it), then the frame groups
                shard_error, hung = rccl_probe(world, rank, dev, {})
            if shard_error is None:
                sp = hotpath.ShardedHotPath(cfg, dev, rank, world, index_prep_mode=args.index_prep,
                                            sparse_exchange=not args.dense_exchange, **rkw)
                if use_rccl:
                    shard_error, hung = rccl_probe(world, rank, dev, sp.exchange.groups)
            if shard_error is None:
                sp_inputs = sp.make_inputs(seed=0)
                for _ in range(2):
                    sp.step(sp_inputs)
                ev = torch.cuda.Event()
                ev.record()
                if not _poll(ev.query, 60.0):
                    shard_error, hung = 'the camera-frame step did not complete within 60 s', True
        except Exception as e:       # noqa: BLE001
            shard_error = f'{type(e).__name__}: {e}'[:300]
        # agreement over the gloo control group only (120 s timeout): never another RCCL call after a failure
        any_hung = not agree(not hung, world)
        if any_hung:
            if rank == 0:
                print(json.dumps({'error': 'a collective of the camera-frame exchange never completed; the device is left with '
                                           'a spinning kernel, so no fallback layout can be timed in this process',
                                  'detail': shard_error}), file=sys.stderr, flush=True)
            os._exit(4)
        if not agree(shard_error is None, world):
            shard_error = shard_error or 'another rank failed in the camera-frame exchange'
            sp, shard = None, 'samples'
    if shard == 'cameras':
        # round 1's layout: rank r < n_cams owns cameras r, r + world', ... (world' active ranks)
        n_active = min(world, cfg.n_cams)
        active = rank < n_active
        my_cams = list(range(rank, cfg.n_cams, n_active)) if active else [0]
        hp = hotpath.HotPath(cfg, dev, cams=my_cams, index_prep_mode=args.index_prep, overlap=not args.no_overlap, **rkw)
    else:
        # whole-sample instance: the N = 1 step, the 'samples' / 'frames' layouts, and every layout's kernel figures
        hp = hotpath.HotPath(cfg, dev, index_prep_mode=args.index_prep, overlap=not args.no_overlap,
                             blend_workgroups='auto' if args.blend_workgroups == 'auto' else int(args.blend_workgroups),
                             lss_pool_backend=None if args.lss_pool == 'auto' else args.lss_pool,
                             gaussians=args.gaussians, **rkw)
    depth, feat = hp.make_inputs(seed=0 if shard in ('cameras', 'camera_frames') else rank)

    step_k = [0]

    def step_whole():
        if hp.alternate:                       # new Gaussian parameters every step (--gaussians objects)
            hp.set_phase(step_k[0] & 1)
            step_k[0] += 1
        out = hp.step(depth, feat)
        if shard == 'frames':
            fused = torch.cat((out[0], out[-2] if cfg.hoa else out[1]), 1)       # (frames, Z*C + C, Y, X)
            sharding.gather_frames(fused, world * fused.shape[0])
        return out

    def step_cameras():
        # pools (+ render of the owned cameras) -> partial fused BEV -> ONE all_reduce -> HOA everywhere
        if active:
            lss, ht = hp.pool_step(depth, feat)
            fused = torch.cat((lss, ht), 1)
            if cfg.render:
                hp.render()
        else:
            fused = torch.zeros(hp.batch, (Z + 1) * cfg.channels, Y, X, device=dev)
        sharding.reduce_partial_bev(fused)
        if cfg.hoa:
            hp.hoa_step(fused[:, Z * cfg.channels:])
        return fused

    step = {'camera_frames': (lambda: sp.step(sp_inputs)), 'cameras': step_cameras}.get(shard, step_whole)
    for _ in range(args.warmup):
        step()
    # device duration of the kernels of interest, HIP events on their launch stream inside the timed region
    t_blend = _lib.KernelTimer(k_blend, 2 * args.steps + 8) if cfg.render else None
    t_pool = _lib.KernelTimer(_lib.K_BEV_POOL_FWD, 2 * (cfg.batch * cfg.n_frames if sp else 1) * args.steps + 8)
    t_mfma = _lib.KernelTimer(_lib.K_BEV_POOL_MFMA, 2 * (cfg.batch * cfg.n_frames if sp else 1) * args.steps + 8)
    t_panel = _lib.KernelTimer(_lib.K_BEV_POOL_PANEL, 2 * (cfg.batch * cfg.n_frames if sp else 1) * args.steps + 8)
    t_cw = _lib.KernelTimer(_lib.K_BEV_POOL_CELL_WEIGHTS, 2 * (cfg.batch * cfg.n_frames if sp else 1) * args.steps + 8)
    for t in (t_blend, t_pool, t_mfma, t_panel, t_cw):
        if t is not None:
            t.arm()
    # K blocks of `steps` steps, each bracketed like the contract says; the reported block is the MEDIAN one
    # (ms_per_step x steps = that block), min / max beside it.  The kernel timers cover the first block.
    blocks = [timed(step, args.steps, world, dev)]
    _lib.KernelTimer.disarm_all()
    n_blocks = args.blocks if args.blocks > 0 else (9 if args.steps <= 50 else 5)
    for _ in range(max(1, n_blocks) - 1):
        blocks.append(timed(step, args.steps, world, dev))
    elapsed = sorted(blocks)[len(blocks) // 2]
    # planned renders with the host guard: ONE status read per plan verifies every render of the warm-up and the timed
    # region (raises if a Gaussian left the plans' extent bound: those renders would not be valid)
    hp.check_render_plans()
    for sub in (sp.subs.values() if sp is not None else ()):
        sub.check_render_plans()

    strong = shard in ('cameras', 'camera_frames')
    # ---- the same kernels alone on the device (renders on the main stream) ------------------------------------
    iso_blend = iso_pool = iso_mfma = iso_panel = iso_cw = None
    if shard in ('none', 'samples', 'frames'):
        was = hp.overlap
        hp.overlap = False
        ib = _lib.KernelTimer(k_blend, 64) if cfg.render else None
        ip = _lib.KernelTimer(_lib.K_BEV_POOL_FWD, 64)
        im = _lib.KernelTimer(_lib.K_BEV_POOL_MFMA, 64)
        ipn, icw = _lib.KernelTimer(_lib.K_BEV_POOL_PANEL, 64), _lib.KernelTimer(_lib.K_BEV_POOL_CELL_WEIGHTS, 64)
        for t in (ib, ip, im, ipn, icw):
            if t is not None:
                t.arm()
        torch.cuda.synchronize()
        for _ in range(min(16, args.steps)):
            hp.step(depth, feat)
        torch.cuda.synchronize()
        _lib.KernelTimer.disarm_all()
        iso_blend = ib.mean_ms() if ib is not None else None
        iso_pool = ip.mean_ms()
        iso_mfma = im.mean_ms()          # None when no pooling ran on the matrix cores
        iso_panel, iso_cw = ipn.read_ms(), icw.mean_ms()      # the panel launches alternate LSS, HT
        im.close(), ipn.close(), icw.close()
        for t in (ib, ip):
            if t is not None:
                t.close()
        hp.overlap = was
    # ---- the step with the index preparation inside (the reference with accelerate=False) ---------------------
    per_step_ms = per_step_devgeom_ms = None
    if shard in ('none', 'samples') and args.index_prep == 'cached' and not args.no_per_step:
        hp2 = hotpath.HotPath(cfg, dev, index_prep_mode='per_step', overlap=not args.no_overlap, **rkw)
        n2 = max(5, min(args.steps, 50))
        for _ in range(3):
            hp2.step(depth, feat)
        per_step_ms = 1e3 * timed_median(lambda: hp2.step(depth, feat), n2, world, dev) / n2
        del hp2
        hp3 = hotpath.HotPath(cfg, dev, index_prep_mode='per_step', overlap=not args.no_overlap, device_geometry=True, **rkw)
        for _ in range(3):
            hp3.step(depth, feat)
        per_step_devgeom_ms = 1e3 * timed_median(lambda: hp3.step(depth, feat), n2, world, dev) / n2
        del hp3
    # ---- the step of a sample whose POSE is new too: nothing calibration- or pose-dependent cached.  Two ways to render
    # such a sample: (a) the per-call pipeline (preprocess + depth buckets + in-LDS sort, no plan at all); (b) the render
    # plan rebuilt on the device inside the step.  Both without a host read anywhere; `per_sample_ms` is the faster. ------
    per_sample = None
    if shard in ('none', 'samples') and cfg.render and not args.no_per_step:
        n2 = max(5, min(args.steps, 50))

        def variant(**kw):
            h = hotpath.HotPath(cfg, dev, index_prep_mode='per_step', overlap=not args.no_overlap, device_geometry=True,
                                render_guard=args.render_guard, **kw)
            for _ in range(3):
                h.step(depth, feat)
            res = {'eager_ms': 1e3 * timed_median(lambda: h.step(depth, feat), n2, world, dev) / n2}
            h.check_render_plans()
            # the same step replayed as ONE hipGraph (nothing in it reads back or allocates)
            try:
                cap_stream = torch.cuda.Stream(dev)
                cap_stream.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(cap_stream):
                    for _ in range(2):
                        h.step(depth, feat)
                torch.cuda.current_stream(dev).wait_stream(cap_stream)
                torch.cuda.synchronize()
                graph = torch.cuda.CUDAGraph()
                with torch.cuda.graph(graph):
                    out_g = h.step(depth, feat)
                torch.cuda.synchronize()
                for _ in range(3):
                    graph.replay()
                res['graph_ms'] = 1e3 * timed(graph.replay, n2, world, dev) / n2
                ref = h.step(depth, feat)
                torch.cuda.synchronize()
                res['graph_outputs_equal_eager'] = bool(torch.equal(ref[0], out_g[0]) and
                                                        torch.equal(ref[2][0]['color'], out_g[2][0]['color']))
                h.check_render_plans()
                del graph, out_g, ref
            except Exception as e:       # noqa: BLE001
                res['graph_ms'], res['graph_outputs_equal_eager'] = None, f'{type(e).__name__}: {e}'[:200]
            return h, res

        h_pc, r_pc = variant(render_mode='per_call')
        del h_pc
        per_sample = {'per_call_render': r_pc}
        if planned:
            h_pl, r_pl = variant(render_mode='planned', plan_rebuild='per_step')
            plans4 = h_pl._plans()

            def rebuild_all():
                for entry in plans4:
                    entry[0].rebuild(entry[3]['cams'])
            for _ in range(3):
                rebuild_all()
            r_pl['render_plan_build_ms'] = 1e3 * timed_median(rebuild_all, n2, world, dev) / n2
            r_pl['plan_records'] = {'kept': [int(sum(e[0].kept)) if e[0].kept else None for e in plans4],
                                    'capacity': [int(e[0].capacity) for e in plans4], 'views': [int(e[0].V) for e in plans4]}
            per_sample['plan_rebuilt_per_step'] = r_pl
            del h_pl, plans4
        per_sample['per_sample_ms'] = min(x for r in list(per_sample.values()) for x in (r['eager_ms'], r.get('graph_ms')) if x)
    # ---- the render on Gaussian sets that are NOT its best case: SURVEY 8d's stress set and an
    # object-centric one, parameters new every step ---------------------------------------------------------------
    gaussian_sets = None
    if shard == 'none' and planned and not args.no_gaussian_sets and not args.no_per_step:
        gaussian_sets = {}
        for kind in ('init', 'stress', 'objects'):
            try:
                gaussian_sets[kind] = gaussian_set_figures(cfg, dev, kind, depth, feat, steps=max(5, min(args.steps, 50)))
            except Exception as e:       # noqa: BLE001
                gaussian_sets[kind] = {'error': f'{type(e).__name__}: {e}'[:300]}
    # ---- weak-scaling secondary of the sharded default: every rank a whole sample ---------------------------
    samples_layout = None
    if shard == 'camera_frames':
        d2, f2 = hp.make_inputs(seed=rank)
        for _ in range(min(args.warmup, 5)):
            hp.step(d2, f2)
        e2 = timed(lambda: hp.step(d2, f2), args.steps, world, dev)
        samples_layout = {'value': hp.bev_voxels_per_step * world * args.steps / e2, 'unit': 'BEV voxels/s',
                          'rendered_views_per_sec': hp.views_per_step * world * args.steps / e2,
                          'ms_per_step': 1e3 * e2 / args.steps, 'scaling': 'weak',
                          'sharding': f'{world} ranks x 1 sample (6 cams x {cfg.n_frames} frames) each, no data-path collective'}

    # ---- the sharded default with its exchange pipelined ACROSS steps (step k's collectives under step k + 1's
    # poolings and renders; outputs one step late): the form of the camera-frame split that is not bound by its own
    # exchange latency.  Timed LAST, behind every other figure of the line, and guarded like the first RCCL steps: RCCL
    # has never run this path on the build box, so its warm-up is waited for by POLLING with a time limit and the ranks
    # agree over gloo before anyone enters the timed region; a collective that never completes costs this one field
    # (the line is still printed, then the processes leave without tearing the communicator down). ------------------
    pipelined_layout = None
    pipelined_hung = False
    if shard == 'camera_frames' and sp is not None:
        perr = None
        try:
            for _ in range(min(max(args.warmup, 2), 5)):
                sp.step_pipelined(sp_inputs)
            sp.flush_pipelined()
            ev = torch.cuda.Event()
            ev.record()
            if not _poll(ev.query, 60.0):
                perr, pipelined_hung = 'the pipelined camera-frame steps did not complete within 60 s', True
        except Exception as e:       # noqa: BLE001
            perr = f'{type(e).__name__}: {e}'[:300]
        pipelined_hung = not agree(not pipelined_hung, world)
        if not agree(perr is None, world) or pipelined_hung:
            pipelined_layout = {'error': perr or 'another rank failed in the pipelined exchange'}
        else:
            def run_pipelined():
                sp.step_pipelined(sp_inputs)
            ep = timed(run_pipelined, args.steps, world, dev)
            sp.flush_pipelined()
            pipelined_layout = {'value': sp.bev_voxels_per_step * args.steps / ep, 'unit': 'BEV voxels/s',
                                'rendered_views_per_sec': cfg.batch * cfg.n_frames * cfg.n_cams * args.steps / ep,
                                'ms_per_step': 1e3 * ep / args.steps, 'scaling': 'strong',
                                'latency_steps': 2,
                                'sharding': 'camera-frames over all ranks as in the headline; step k\'s reduce_scatter + '
                                            'all_gather run on a communication stream under step k + 1\'s poolings and '
                                            'renders (two buffer sets); HOA-3 and the outputs of a step come one call late'}
    if rank == 0:
        blend_ms = t_blend.mean_ms() if t_blend is not None else None
        pool_ms = t_pool.mean_ms()
        d_numel, f_numel = depth.numel(), feat.numel()
        # one entry per pooling, each with its OWN algorithmic bytes and its own in-step launch duration: with the
        # default backends the LSS ranks run the tile kernel (K_BEV_POOL_FWD), the HT ranks the MFMA panels (K_BEV_POOL_MFMA)
        mfma_ms = t_mfma.mean_ms()
        lss_on_mfma = getattr(hp, 'lss_pool_backend', 'tile') == 'mfma'
        ht_on_mfma = getattr(hp, 'ht_pool_backend', 'tile') == 'mfma' and mfma_ms
        # the panel latency kernel (the default where nothing VALU-bound runs beside the poolings): its timer holds the LSS
        # and the HT launch of every step in turn; the shared weight pre-pass has a timer of its own
        panel_all = t_panel.read_ms()
        on_panel = {nm: getattr(hp, nm + '_pool_backend', 'tile') == 'panel' and len(panel_all) > 0 for nm in ('lss', 'ht')}
        both_panel = on_panel['lss'] and on_panel['ht']

        def mean(v):
            return sum(v) / len(v) if v else None

        def panel_ms(which, series):
            if not series:
                return None
            if both_panel:
                return mean(series[0::2] if which == 'lss' else series[1::2])
            return mean(series)

        def pool_entry(name, plan, kernel, in_step_ms, iso_ms, counters):
            alg = plan.algorithmic_bytes(d_numel, f_numel)
            return {'pooling': name, 'kernel': kernel, 'bound': 'hbm', 'algorithmic_bytes_per_launch': alg,
                    'avg_launch_us': 1e3 * in_step_ms if in_step_ms else None,
                    'achieved': alg / (in_step_ms * 1e-3) / 1e9 if in_step_ms else None, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                    'frac': alg / (in_step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if in_step_ms else None,
                    'isolated_avg_launch_us': 1e3 * iso_ms if iso_ms else None,
                    'isolated_frac': alg / (iso_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if iso_ms else None,
                    'traffic': hbm_traffic(counters), 'traffic_source': PMC_FILE if counters else None}
        tile_c, mfma_c = pmc_counters('bev_pool_tile_kernel', args), pmc_counters('bev_pool_mfma_kernel', args)
        both_tile = not lss_on_mfma and not ht_on_mfma      # one timer, two launches per step: the mean of both
        pools = {'bound': 'hbm', 'unit': 'GB/s', 'peak': HBM_PEAK_GBS,
                 'algorithmic_bytes_formula': '4*(B*N*D*H*W + B*N*H*W*C + 3*Np + 2*Nv + B*Z*Y*X*C) per pooling (SURVEY 8d)',
                 'lss': (pool_entry('LSS (frustum ranks)', hp.lss, t_panel.kernel_name, panel_ms('lss', panel_all),
                                    panel_ms('lss', iso_panel), pmc_counters('bev_pool_panel_kernel', args)) if on_panel['lss'] else
                         pool_entry('LSS (frustum ranks)', hp.lss, 'bev_pool_mfma_kernel<*>' if lss_on_mfma else t_pool.kernel_name,
                                    mfma_ms if lss_on_mfma else pool_ms, iso_mfma if lss_on_mfma else iso_pool,
                                    mfma_c if lss_on_mfma else tile_c)),
                 'ht': (pool_entry('HT (height-sampling ranks)', hp.ht, t_panel.kernel_name, panel_ms('ht', panel_all),
                                   panel_ms('ht', iso_panel), pmc_counters('bev_pool_panel_kernel', args)) if on_panel['ht'] else
                        pool_entry('HT (height-sampling ranks)', hp.ht, 'bev_pool_mfma_kernel<*>' if ht_on_mfma else t_pool.kernel_name,
                                   mfma_ms if ht_on_mfma else pool_ms, iso_mfma if ht_on_mfma else iso_pool,
                                   mfma_c if ht_on_mfma else tile_c)),
                 'cell_weights_prepass': ({'kernel': t_cw.kernel_name, 'avg_launch_us': 1e3 * t_cw.mean_ms(),
                                           'isolated_avg_launch_us': 1e3 * iso_cw if iso_cw else None,
                                           'note': 'one launch per step sums the cell weights of BOTH panel plans (they read '
                                                   'the same depth tensor); the panel poolings above start from them'}
                                          if (on_panel['lss'] or on_panel['ht']) and t_cw.mean_ms() else None),
                 'launches_timed': {'tile': t_pool.count(), 'mfma': t_mfma.count(), 'panel': len(panel_all)},
                 'note': ('both poolings on the tile kernel: its timer holds both launches of a step, the durations above '
                          'are their mean' if both_tile else
                          'a kernel timer per backend: the durations are those of the named pooling alone '
                          '(both on one backend: their mean)')}
        # (kept for readers of the earlier rounds' lines: the LSS launch under the old top-level keys)
        pools.update({k: pools['lss'][k] for k in ('kernel', 'avg_launch_us', 'achieved', 'frac', 'traffic',
                                                     'isolated_avg_launch_us', 'isolated_frac', 'algorithmic_bytes_per_launch')})
        if iso_mfma:
            # the HT pooling on the matrix cores: duration alone on the chip, MFMA counters of the PMC file
            mc = pmc_counters('bev_pool_mfma_kernel', args)
            simd_cycles = 1024 * iso_mfma * 1e-3 * CLOCK_HZ
            pools['mfma'] = {'kernel': 'bev_pool_mfma_kernel<*>', 'pooling': 'HT (height-sampling ranks)',
                             'isolated_avg_launch_us': 1e3 * iso_mfma,
                             'SQ_INSTS_VALU_MFMA_MOPS_F32': mc.get('SQ_INSTS_VALU_MFMA_MOPS_F32'),
                             'SQ_VALU_MFMA_BUSY_CYCLES': mc.get('SQ_VALU_MFMA_BUSY_CYCLES'),
                             'mfma_busy_frac': (mc['SQ_VALU_MFMA_BUSY_CYCLES'] / simd_cycles
                                                if mc.get('SQ_VALU_MFMA_BUSY_CYCLES') else None),
                             'traffic': hbm_traffic(mc), 'source': PMC_FILE if mc else None,
                             'note': 'per 8x8-voxel tile out[64xC] = W[64xR].F[RxC] on v_mfma_f32_16x16x4_f32; busy = '
                                     'SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x kernel cycles at 2.4 GHz): the matrix pipes '
                                     'are mostly idle, a panel is a latency chain (rows, depth gathers, two barriers)'}
        if (sp is None and shard == 'none' and hp.lss.n_points and hp.ht.n_points and cfg.channels in (64, 80, 96, 128)
                and not args.no_pool_backends):
            # the three pooling kernels alone on the device, back-to-back launches between two events (no kernel timer:
            # these launches include their launch gap, ~1 us over the kernel alone at these sizes)
            try:
                def alone(fn, n=30):
                    for _ in range(3):
                        fn()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    torch.cuda.synchronize()
                    e0.record()
                    for _ in range(n):
                        fn()
                    e1.record()
                    torch.cuda.synchronize()
                    return 1e3 * e0.elapsed_time(e1) / n
                alt = {}
                pplans = {}
                for nm, pl in (('lss', hp.lss), ('ht', hp.ht)):
                    dplan = pl.device_plan or bevpool.DevicePoolPlan(pl.ranks_depth, pl.ranks_feat, pl.ranks_bev, pl.bev_shape,
                                                                     pl.starts, pl.lengths)
                    mplan = bevpool.MfmaPoolPlan(pl.ranks_depth, pl.ranks_feat, pl.ranks_bev, pl.bev_shape,
                                                 group=8 if nm == 'ht' else 2)
                    pplan = pplans[nm] = bevpool.MfmaPoolPlan(pl.ranks_depth, pl.ranks_feat, pl.ranks_bev, pl.bev_shape, group=8,
                                                              unit_cost=None if nm == 'ht' else 8.0)
                    alt[nm] = {'tile': alone(lambda: bevpool.bev_pool_v2_planned(depth, feat, dplan)),
                               'mfma': alone(lambda: bevpool.bev_pool_v2_mfma(depth, feat, mplan)),
                               'panel': alone(lambda: bevpool.bev_pool_v2_panel(depth, feat, pplan, weights_ready=True))}
                    del mplan

                def both_panel():
                    bevpool.bev_pool_cell_weights(depth, pplans['lss'], pplans['ht'])
                    bevpool.bev_pool_v2_panel(depth, feat, pplans['lss'], weights_ready=True)
                    bevpool.bev_pool_v2_panel(depth, feat, pplans['ht'], weights_ready=True)
                alt['panel_weight_prepass_both_plans'] = alone(lambda: bevpool.bev_pool_cell_weights(depth, pplans['lss'], pplans['ht']))
                alt['both_poolings_panel'] = alone(both_panel)
                alt['both_poolings_as_in_step'] = alone(lambda: hp.pool_step(depth, feat))
                alt['note'] = ('us per call, each kernel alone on the device: tile = bev_pool_tile_kernel, mfma = bev_pool_mfma_kernel, '
                               'panel = bev_pool_panel_kernel with its cell weights ready (+ the pre-pass, one launch for both '
                               'plans).  The step uses ' + str(getattr(hp, 'lss_pool_backend', 'tile')) + ' (LSS) / ' +
                               str(getattr(hp, 'ht_pool_backend', 'tile')) + ' (HT): beside the persistent blend the panel form '
                               'ties with them (DESIGN.md 4.1c)')
                pools['backends_alone_us'] = alt
                del pplans
            except Exception as e:      # a diagnostic must not cost the line
                pools['backends_alone_us'] = {'error': repr(e)}
        if sp is not None:
            pools['note'] = ('camera-sharded pools: each launch pools this rank\'s cameras of one frame into a full-size '
                             'partial grid; algorithmic bytes are those of the whole-sample pools, for reference only')
        if cfg.render:
            # the blend is VALU-bound: flops = 20 per pixel.record, pixel.records counted as the
            # contributor index every pixel stopped at (a lower bound of what the kernel evaluates)
            n_launch = len(hp._plans()) if planned else hp.batch      # a planned launch blends every frame of its plan
            evals = float(sum(int(o['n_contrib'].sum().item()) for o in hp.render(want_n_contrib=True))) / n_launch
            # (the FIRST pass of the planned blend: the second — usually an empty launch — is a symbol of its own)
            c = pmc_counters('raster_blend_sorted_kernel<true, false, false, false>' if planned else
                             'raster_blend_kernel<false, false, true, false>', args)
            cycles = blend_ms * 1e-3 * CLOCK_HZ
            # The planned kernel's OWN count: a wave of it evaluates only the records whose ellipse reaches its 16 x 8
            # pixel block (fewer than a tile's list up to the stop index, which is what n_contrib counts): the
            # instrumented build of the same kernel (one extra launch, after the timed region) reports the wave-records
            # each wave evaluated; x 128 pixels.  The roofline prices THAT work; the reference count stays beside it.
            reference_evals, executed = evals, None
            if planned and sp is None:
                try:
                    Himg, Wimg = cfg.input_size
                    tiles = ((Wimg + 15) // 16) * ((((Himg + 15) // 16) + 1) // 2)
                    executed = 0.0
                    for entry in hp._plans():
                        n_wg = tiles * entry[2] * len(hp.cams)
                        buf = torch.zeros(n_wg * 4 * 8, dtype=torch.int64, device=dev)
                        torch.cuda.synchronize()
                        _lib.lib().ocrf_diag_plan_stats(_lib.ptr(buf))
                        try:
                            hp._render_planned(entry)
                            torch.cuda.synchronize()
                        finally:
                            _lib.lib().ocrf_diag_plan_stats(None)
                        executed += float(buf.view(n_wg, 4, 8)[:, :, 6].sum().item()) * 128.0
                    executed /= n_launch
                    evals = executed
                except Exception:       # noqa: BLE001
                    executed = None
            tfl = BLEND_FLOPS_PER_PIXEL_RECORD * evals / (blend_ms * 1e-3) / 1e12
            traffic = hbm_traffic(c)
            grid_note = None
            if planned and getattr(hp, 'overlap', False) and getattr(hp, 'blend_workgroups', None) == 'auto':
                cus = torch.cuda.get_device_properties(dev).multi_processor_count
                grid_note = {'workgroups_timed_region': 11 * cus // 4, 'workgroups_isolated': _lib.lib().ocrf_diag_plan_resident(),
                             'note': 'beside the pooling / HOA stream the persistent blend is launched on 2.75 workgroups '
                                     'per CU of the five the chip holds (DESIGN 5); "frac" prices that partial-occupancy '
                                     'launch against the whole chip\'s peak, "isolated" is the full grid'}
            # which resource bounds the kernel is read from the counters, not assumed: VALU busy share of the
            # SIMDs over the kernel's span vs the share of wave cycles spent waiting; without counters of THIS build the
            # round-5 finding stands (0.49 busy, 0.59 waiting: a latency / occupancy kernel priced against the VALU peak)
            cyc_b = (iso_blend or blend_ms) * 1e-3 * CLOCK_HZ
            valu_busy = (4.0 * c['SQ_ACTIVE_INST_VALU'] / (N_SIMD * cyc_b)) if c.get('SQ_ACTIVE_INST_VALU') else None
            wait_share = (c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES']) if c.get('SQ_WAIT_ANY') and c.get('SQ_WAVE_CYCLES') else None
            bound = 'valu' if (valu_busy is not None and valu_busy >= 0.6) else 'latency'
            roofline = {
                'bound': bound,
                'bound_basis': {'valu_busy_frac': valu_busy, 'wave_wait_share': wait_share,
                                'rule': 'valu if the SIMDs issue VALU work >= 0.6 of the kernel\'s span (PMC of this build), else '
                                        'latency: achieved / peak / frac still price the executed flops against the fp32 '
                                        'vector peak', 'source': PMC_FILE if valu_busy is not None else
                                'no counters of this build: round 5 measured 0.49 busy / 0.59 waiting'},
                'kernel': t_blend.kernel_name,
                'achieved': tfl, 'peak': FP32_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': tfl / FP32_PEAK_TFLOPS,
                'flops_per_pixel_record': BLEND_FLOPS_PER_PIXEL_RECORD, 'pixel_records_per_launch': evals,
                'pixel_records_source': ('executed: wave-records the planned blend evaluated (its instrumented build, one launch '
                                         'after the timed region) x 128 pixels' if executed is not None else
                                         'reference: sum over pixels of the contributor index they stopped at (n_contrib)'),
                'reference_pixel_records_per_launch': reference_evals,
                'avg_launch_us': 1e3 * blend_ms, 'launches_timed': t_blend.count(),
                'grid': grid_note,
                'traffic': traffic, 'traffic_source': (PMC_FILE + ' (' + PMC_META + ' matches this run)') if traffic else None,
                'hbm': ({'bytes_per_launch_pmc': traffic, 'achieved_GBs': traffic / (blend_ms * 1e-3) / 1e9,
                         'frac_of_peak': traffic / (blend_ms * 1e-3) / 1e9 / HBM_PEAK_GBS} if traffic else None),
                'isolated': ({'avg_launch_us': 1e3 * iso_blend,
                              'achieved': BLEND_FLOPS_PER_PIXEL_RECORD * evals / (iso_blend * 1e-3) / 1e12,
                              'frac': BLEND_FLOPS_PER_PIXEL_RECORD * evals / (iso_blend * 1e-3) / 1e12 / FP32_PEAK_TFLOPS,
                              'note': 'same kernel, stages not overlapped (after the timed region)'} if iso_blend else None)}
            if c.get('SQ_INSTS_VALU'):
                # counters are per launch of the --no-overlap run -> priced against the isolated duration
                cyc = (iso_blend or blend_ms) * 1e-3 * CLOCK_HZ
                roofline['valu'] = {
                    'source': PMC_FILE, 'SQ_INSTS_VALU': c['SQ_INSTS_VALU'], 'SQ_INSTS_SALU': c.get('SQ_INSTS_SALU'),
                    'SQ_ACTIVE_INST_VALU_quadcycles': c.get('SQ_ACTIVE_INST_VALU'),
                    'valu_insts_per_pixel_record_pair': c['SQ_INSTS_VALU'] * 64 / (evals / 2) if evals else None,
                    'issue_frac': c['SQ_INSTS_VALU'] * VALU_CYCLES_PER_INST / (N_SIMD * cyc),
                    'busy_frac': (4.0 * c['SQ_ACTIVE_INST_VALU'] / (N_SIMD * cyc)) if c.get('SQ_ACTIVE_INST_VALU') else None,
                    'salu_per_valu': (c['SQ_INSTS_SALU'] / c['SQ_INSTS_VALU']) if c.get('SQ_INSTS_SALU') else None,
                    'note': 'issue_frac = wave64 VALU instructions x 2 cycles (SIMD-32) / (1024 SIMDs x kernel cycles at '
                            '2.4 GHz); busy_frac = SQ_ACTIVE_INST_VALU (quad-cycles) x 4 / the same denominator'}
            del cycles
        else:
            roofline = dict(pools)
        voxels = hp.bev_voxels_per_step * (1 if strong else world) * args.steps
        views = (cfg.batch * cfg.n_frames * cfg.n_cams if cfg.render else 0) * (1 if strong else world) * args.steps
        if shard == 'camera_frames':
            ex = sp.exchange
            sharding_desc = {
                'layout': sp.plan.describe(), 'rccl_ranks': world, 'idle_ranks': sp.plan.idle_ranks,
                'collectives_per_step': ((['wedge-sparse step 1: batched isend / irecv of the touched strips of each member\'s plane '
                                           'block inside a frame group, added in member order'] if ex.touched else
                                          ['reduce_scatter(sum) of the (padded) partial fused grid inside each frame group'])
                                         if ex.partial else []) + ['all_gather of the finished plane blocks over the world'],
                'bytes_received_per_rank_reduce_scatter': ex.bytes_reduce_scatter,
                'bytes_received_per_rank_reduce_scatter_dense_form': ex.bytes_reduce_scatter_dense,
                'bytes_received_per_rank_all_gather': ex.bytes_all_gather,
                'in_place_all_gather': bool(ex.direct), 'overlap': 'collectives asynchronous beside the renders (side HIP '
                'stream) and HOA-1/2',
                'hoa': 'sharded by frame: a rank runs HOA-1/2 only for the frames it has a part in (' +
                       str(len(sp.my_frames)) + ' of ' + str(sp.n_frames) + ' here) and HOA-3 in place on its own plane '
                       'blocks between the two collectives (sharding.gate_blocks: per-block channel statistics, one small '
                       'all_gather inside the frame group); the world all_gather carries LSS planes + GATED HT planes + '
                       'the opacity BEV plane'}
        else:
            sharding_desc = {'none': 'none',
                             'samples': f'{world} ranks x 1 sample (6 cams x {cfg.n_frames} frames) each, no data-path collective',
                             'frames': f'{world} x {cfg.n_frames} frames of one sequence, one RCCL all_gather of the fused BEV per step',
                             'cameras': f'1 sample, cameras over {min(world, cfg.n_cams)} of {world} ranks, one RCCL all_reduce of the fused BEV per step'}[shard]
        plan_note = 'one plan per frame'
        plans_built = getattr(hp, 'render_plans', None)
        if planned and plans_built:
            plan_note = (f'{len(plans_built)} plan(s) for {hp.batch} frame(s): frames that fit a 32-view plan share one '
                         'head + one blend launch')
        out = {
            'metric': 'BEV voxels/sec + rendered views/sec, 6-cam 256x704',
            'value': voxels / elapsed, 'unit': 'BEV voxels/s',
            'rendered_views_per_sec': views / elapsed,
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / args.steps,
            'ms_per_step_blocks': {'n': len(blocks), 'median': 1e3 * elapsed / args.steps,
                                   'min': 1e3 * min(blocks) / args.steps, 'max': 1e3 * max(blocks) / args.steps,
                                   'in_order': [round(1e3 * b / args.steps, 5) for b in blocks]},
            'per_step_ms': per_step_ms,
            'per_step_device_geometry_ms': per_step_devgeom_ms,
            # a sample with a NEW POSE (the reference's only working mode: cameras per sample, accelerate=False): rank
            # vectors by the device index preparation, renders per call or from a plan rebuilt on the device, every step
            'per_sample_ms': per_sample['per_sample_ms'] if per_sample else None,
            'render_plan_build_ms': (per_sample.get('plan_rebuilt_per_step', {}).get('render_plan_build_ms')
                                     if per_sample else None),
            'per_sample': (None if per_sample is None else {
                **per_sample,
                'what': 'HotPath(index_prep_mode="per_step", device_geometry=True, ...).step: nothing calibration- or '
                        'pose-dependent is cached and nothing is read back; per_call_render = preprocess + depth buckets + '
                        'in-LDS sort every render (no plan); plan_rebuilt_per_step = RasterPlan.rebuild (classify -> scan -> '
                        'records -> one radix sort -> gather) + update + sorted blend; graph_ms = the same step captured once '
                        'and replayed as one hipGraph; per_sample_ms = the fastest of the four',
                'amortisation': 'ms_per_step (headline) builds plan and rank vectors once per calibration; '
                                'per_sample_ms - ms_per_step is what a per-sample calibration adds; a plan pays off after '
                                'render_plan_build_ms / (per-call - planned render time) renders of one pose',
                'reference': 'view_transformer_ocrf.py:1140-1152, detectors/ocrfdet.py:215-223'}),
            'higher_is_better': True,
            **({'sharding_fallback': 'camera_frames -> samples: ' + shard_error} if shard_error else {}),
            'scaling': 'strong' if strong else 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': cfg.name, 'cams': cfg.n_cams, 'frames_per_gpu': cfg.n_frames,
                       'bev': list(cfg.bev_xyz), 'channels': cfg.channels, 'depth_bins': cfg.D,
                       'stages': 'lss_pool+ht_pool' + ('+render' if cfg.render else '') + ('+hoa' if cfg.hoa else ''),
                       'views_per_step': hp.views_per_step if sp is None else cfg.batch * cfg.n_frames * cfg.n_cams,
                       'render_camera': getattr(hp, 'render_convention', None),
                       'render_front_end': (None if not cfg.render else
                                            ('static render plan (cull, depth order, projected centres cached per calibration like the rank '
                                             'vectors; Gaussian means = the fixed voxel grid; ' + plan_note + '); extent bound '
                                             + ('verified by one status read after the timed region' if args.render_guard == 'host'
                                                else 'guarded on the device (per-call pipeline armed behind every render)'))
                                            if planned else 'per call (preprocess + depth-bucket scatter every render)'),
                       'frames': 'each frame its own ego pose and Gaussian parameters (synthetic.ego_motion)',
                       'streams': ('main: pools, then HOA; side HIP stream: per plan the head of every view\'s list, the blend, '
                                   'the extent check; the blend runs as a persistent grid of 2.75 workgroups per CU so that the '
                                   'latency-bound kernels of the main stream keep wave slots (with several blends per step the '
                                   'rest of the grid joins once the main chain is done; DESIGN.md section 5)'
                                   if hp.overlap and cfg.render and planned
                                   else ('main: HOA-1/2, pools, HOA-3; side HIP stream: renders' if hp.overlap and cfg.render
                                         else 'single stream')),
                       'ht_pool': getattr(hp, 'ht_pool_backend', None), 'lss_pool': getattr(hp, 'lss_pool_backend', None),
                       # north_star asks for MFMA on the depth x feature outer product: the MFMA form of both poolings exists
                       # (csrc/bev_pool_mfma.hip) and is timed alone below (pools.backends_alone_us.*.mfma); the default step
                       # does NOT launch it — the panel form of the same plan is faster alone and in the step
                       'pools_on_mfma': bool(lss_on_mfma or ht_on_mfma),
                       'issue': ('one host call per step (ocrf_hotpath_step: the step\'s library calls recorded once, replayed '
                                 'from C)' if getattr(hp, '_compiled', None) else 'call by call from Python'),
                       'index_prep': 'cached (accelerate=True semantics); per_step_ms = the same step with the HIP index '
                                     'preparation inside (accelerate=False semantics, the reference\'s working mode), calibration '
                                     'algebra on the host as the reference\'s own torch calls (rank vectors bit-exact); '
                                     'per_step_device_geometry_ms = the same with that algebra on the GPU too'
                                     if args.index_prep == 'cached' else 'per step, HIP (accelerate=False semantics)',
                       'sharding': sharding_desc},
            'roofline': roofline,
        }
        if cfg.render:
            out['pools'] = pools
        if gaussian_sets is not None:
            out['gaussian_sets'] = gaussian_sets
        out['config']['gaussians'] = getattr(hp, 'gaussians', None)
        if samples_layout is not None:
            out['samples_layout'] = samples_layout
        if pipelined_layout is not None:
            out['pipelined_camera_frames'] = pipelined_layout
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(hp, depth, feat, args.cpu_seconds)
            out['gpu_over_cpu'] = out['value'] / out['cpu_baseline']['value']
        emit(out)
    for t in (t_blend, t_pool, t_mfma, t_panel, t_cw):
        if t is not None:
            t.close()
    if world > 1:
        import torch.distributed as dist
        if shard_error or pipelined_hung:
            # a communicator that failed above may never shut down: leave without the teardown (the line is printed)
            sys.stdout.flush()
            os._exit(0)
        try:
            dist.destroy_process_group()
        except Exception:       # noqa: BLE001
            pass


if __name__ == '__main__':
    main()
