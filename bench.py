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
  samples (default) — every rank owns whole samples (6 cameras x n_frames each), the layout of the
      reference's own multi-GPU runs (tools/dist_test.sh -> MMDistributedDataParallel, one process
      per GPU).  The hot path has no exchange step in this layout, so there is no data-path
      collective; weak scaling.
  frames  — every rank owns n_frames frames of ONE (world x n_frames)-frame sequence and one RCCL
      all_gather per step hands every rank the fused BEV of all frames (the operand of the channel
      concat, detectors/ocrfdet.py:274); weak scaling.
  cameras — ONE sample, cameras split over min(world, n_cams) ranks (BASELINE.json configs[3]), one
      RCCL all_reduce(sum) of the partial fused BEV per step, HOA replicated; strong scaling.
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
DEFAULT_CONFIG = 'cfg2_6cam_2frame_bev200x200_render_hoa'


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--config', default=DEFAULT_CONFIG)
    ap.add_argument('--cpu-seconds', type=float, default=12.0, help='CPU-baseline sample budget')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--shard', choices=('samples', 'frames', 'cameras'), default='samples')
    ap.add_argument('--backend', default='nccl', help='torch.distributed backend (nccl = RCCL); gloo only for plumbing tests')
    ap.add_argument('--no-overlap', action='store_true',
                    help='run the renders on the main stream instead of beside the pools + HOA on a side HIP stream')
    ap.add_argument('--no-graph', action='store_true',
                    help="--scope neck: issue the step kernel by kernel instead of replaying it as one hipGraph")
    ap.add_argument('--scope', choices=('hotpath', 'neck'), default='hotpath',
                    help="'hotpath': pools (+ one rendered view per camera + HOA) — the headline step; 'neck': the whole "
                         "OcRFViewTransformerFull.view_transform (one rendered view per sample, like the reference)")
    ap.add_argument('--host-calibration', action='store_true',
                    help="--scope neck --index-prep per_step: hand the calibration tensors over as HOST tensors (the "
                         "dataloader's copies) — the forward then has no device -> host read-back at all")
    ap.add_argument('--index-prep', choices=('cached', 'per_step'), default='cached',
                    help="'per_step': rank vectors recomputed by the HIP index preparation inside every step "
                         "(the reference with accelerate=False); 'cached': once per calibration (accelerate=True)")
    return ap.parse_args()


def cpu_baseline(hp, depth, feat, budget_s):
    """Times the C/OpenMP oracle on a bounded sample of the same step: the two pools of the whole
    step (a few repetitions) and ONE rendered view (the reference-structured rasteriser sorts all
    tile instances, so a full step of views would take minutes); the step time is assembled as
    t_pools + views_per_step * t_view."""
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
    pools()                                                          # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        pools()
        n += 1
        el = time.perf_counter() - t0
        if el >= min(budget_s, 5.0) or n >= 200:
            break
    t_pools = el / n
    t_view, n_views_timed, rendered = 0.0, 0, None
    if hp.cfg.render:
        g = hp.gauss
        Himg, Wimg = hp.cfg.input_size
        args = (hp.voxel_xyz[0].reshape(-1, 3).cpu().numpy(), g['rgb'].cpu().numpy(), g['opacity'].cpu().numpy(),
                g['scales'].cpu().numpy(), g['rotations'].cpu().numpy())
        t0 = time.perf_counter()
        while True:
            v = n_views_timed % len(hp.cams)
            r = oracle.rasterize_forward(*args, hp.render_cams['vm'][v].cpu().numpy(), hp.render_cams['pm'][v].cpu().numpy(),
                                         hp.render_cams['tfx'][v], hp.render_cams['tfy'][v], Himg, Wimg,
                                         np.zeros(3, np.float32))
            rendered = r['num_rendered']
            n_views_timed += 1
            el = time.perf_counter() - t0
            if el >= budget_s or n_views_timed >= 12:
                break
        t_view = el / n_views_timed
    t_step = t_pools + hp.views_per_step * t_view
    return dict(value=hp.bev_voxels_per_step / t_step, unit='BEV voxels/s', cores=oracle.num_threads(),
                kind='port', ms_per_step=1e3 * t_step, ms_pools=1e3 * t_pools, ms_per_view=1e3 * t_view,
                views_per_sec=(hp.views_per_step / t_step) if hp.views_per_step else 0.0,
                sample=f'{n} x (LSS pool + HT pool of the whole step, same inputs and ranks as the GPU)'
                       + (f' and {n_views_timed} rendered view(s) of frame 0 ({rendered} tile instances in the last), '
                          f'step time assembled as pools + {hp.views_per_step} x view' if hp.cfg.render else '')
                       + '; HOA (small torch convs) not included; C/OpenMP oracle')


def bench_neck(args, cfg, dev, world, rank):
    """--scope neck: one step = pre-filter + the whole view_transform of B = n_frames samples per rank
    (whole samples per rank, no data-path collective).  Roofline: the LSS/HT pooling kernel."""
    import torch.distributed as dist
    from ocrfdet_amd import _lib, hotpath
    neck = hotpath.NeckPath(cfg, dev, accelerate=args.index_prep == 'cached', seed=rank,
                            host_calibration=args.host_calibration)
    graphed = args.index_prep == 'cached' and not args.no_graph
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
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
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
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
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
                          ('handed over as host tensors' if args.host_calibration else 'read back from the device (one packed copy)'),
                          'sharding': 'none' if world == 1 else f'{world} ranks x whole samples, no data-path collective'},
               'roofline': {'bound': 'hbm', 'kernel': timer.kernel_name, 'achieved': achieved, 'peak': HBM_PEAK_GBS,
                            'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS if achieved else None, 'traffic': None,
                            'algorithmic_bytes_per_launch': alg, 'avg_launch_us': 1e3 * avg_ms, 'launches_timed': len(ms),
                            'note': 'kernel timed over an eager run of the same step right after the timed region '
                                    '(HIP events cannot bracket a launch inside a graph replay)'},
               'cpu_baseline': None}
        print(json.dumps(out), flush=True)
    timer.close()
    if world > 1:
        dist.destroy_process_group()


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    single_dev = os.environ.get('OCRF_BENCH_SINGLE_DEVICE') == '1'      # plumbing test: all ranks on cuda:0 (gloo)
    if single_dev:
        local_rank = 0
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
        else:
            dist.init_process_group(args.backend)
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)

    from ocrfdet_amd import _lib, hotpath, sharding, synthetic
    _lib.lib()                      # raises if libocrf_hip.so is missing — no fallback
    cfg = synthetic.CONFIGS[args.config]
    shard = args.shard if world > 1 else 'none'
    active = True
    if args.scope == 'neck':
        return bench_neck(args, cfg, dev, world, rank)
    if shard == 'cameras':
        # strong scaling of ONE sample: rank r < n_cams owns cameras r, r + world', ... (world' active ranks)
        n_active = min(world, cfg.n_cams)
        active = rank < n_active
        my_cams = list(range(rank, cfg.n_cams, n_active)) if active else [0]
        hp = hotpath.HotPath(cfg, dev, cams=my_cams, index_prep_mode=args.index_prep, overlap=not args.no_overlap)
    else:
        hp = hotpath.HotPath(cfg, dev, index_prep_mode=args.index_prep, overlap=not args.no_overlap)
    depth, feat = hp.make_inputs(seed=0 if shard == 'cameras' else rank)
    X, Y, Z = cfg.bev_xyz

    def step():
        if shard == 'cameras':
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
        out = hp.step(depth, feat)
        if shard == 'frames':
            fused = torch.cat((out[0], out[-2] if cfg.hoa else out[1]), 1)       # (frames, Z*C + C, Y, X)
            sharding.gather_frames(fused, world * fused.shape[0])
        return out

    for _ in range(args.warmup):
        step()
    dom_id = _lib.K_RASTER_BLEND if cfg.render else _lib.K_BEV_POOL_FWD
    timer = _lib.KernelTimer(dom_id, 2 * args.steps)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    timer.arm()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    timer.disarm()
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # the same kernel alone on the device (renders on the main stream): with the stages overlapped on
    # several HIP streams the in-region duration above includes the time the kernel shares the CUs
    iso_ms = None
    if hp.overlap and cfg.render and shard != 'cameras':
        hp.overlap = False
        iso = _lib.KernelTimer(dom_id, 64)
        torch.cuda.synchronize()
        iso.arm()
        for _ in range(min(16, args.steps)):
            step()
        torch.cuda.synchronize()
        iso.disarm()
        v = iso.read_ms()
        iso_ms = sum(v) / max(len(v), 1)
        iso.close()
        hp.overlap = True
    if rank == 0:
        ms = timer.read_ms()
        avg_ms = sum(ms) / max(len(ms), 1)
        if cfg.render:
            # SURVEY 8(d) per-view figure for what one blend launch touches: 8 B rect + 8 B record
            # per visible Gaussian scanned at least once, 44 B payload per blended record is bounded
            # by the same count, 20 B per pixel written (colour 12, depth 4, T 4) + 4 B n_contrib
            H, W = cfg.input_size
            P = hp.voxel_xyz.shape[1] * hp.voxel_xyz.shape[2]
            alg_bytes = float(len(hp.cams) * (60 * P + 24 * H * W))
            # the blend is VALU-bound: pixel.record evaluations per launch, counted as the
            # contributor index each pixel stopped at (a lower bound of what the kernel evaluates)
            evals = float(hp.render(want_n_contrib=True)[0]['n_contrib'].sum().item())
        else:
            alg_bytes = 0.5 * (hp.lss.algorithmic_bytes(depth.numel(), feat.numel()) +
                               hp.ht.algorithmic_bytes(depth.numel(), feat.numel()))
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        # HBM bytes per launch from the PMC passes committed under profiles/ (rocprofv3 --pmc
        # FETCH_SIZE / WRITE_SIZE in separate runs of this same command, gfx950 x2 fetch correction);
        # bench.py cannot run the profiler on itself, so this is the last committed measurement
        traffic, traffic_src = None, None
        pmc = os.path.join(ROOT, 'profiles', 'r1_pmc_traffic.json')
        if args.config == DEFAULT_CONFIG and os.path.exists(pmc):
            rec = json.load(open(pmc)).get(timer.kernel_name.split('<')[0])
            if rec:
                traffic, traffic_src = rec['hbm_bytes_corrected'], 'profiles/r1_pmc_traffic.json'
        strong = shard == 'cameras'
        voxels = hp.bev_voxels_per_step * (1 if strong else world) * args.steps
        out = {
            'metric': 'BEV voxels/sec + rendered views/sec, 6-cam 256x704',
            'value': voxels / elapsed, 'unit': 'BEV voxels/s',
            'rendered_views_per_sec': (cfg.batch * cfg.n_frames * cfg.n_cams if cfg.render else 0) *
                                      (1 if strong else world) * args.steps / elapsed,
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True,
            'scaling': 'strong' if strong else 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': cfg.name, 'cams': cfg.n_cams, 'frames_per_gpu': cfg.n_frames,
                       'bev': list(cfg.bev_xyz), 'channels': cfg.channels, 'depth_bins': cfg.D,
                       'stages': 'lss_pool+ht_pool' + ('+render' if cfg.render else '') + ('+hoa' if cfg.hoa else ''),
                       'views_per_step': hp.views_per_step, 'render_camera': getattr(hp, 'render_convention', None),
                       'streams': ('main: pools + HOA; side HIP stream: renders' if hp.overlap and cfg.render
                                   else 'single stream'),
                       'index_prep': 'cached (accelerate=True semantics)' if args.index_prep == 'cached' else
                                     'per step, HIP (accelerate=False semantics)',
                       'sharding': {'none': 'none',
                                    'samples': f'{world} ranks x 1 sample (6 cams x {cfg.n_frames} frames) each, no data-path collective',
                                    'frames': f'{world} x {cfg.n_frames} frames of one sequence, one RCCL all_gather of the fused BEV per step',
                                    'cameras': f'1 sample, cameras over {min(world, cfg.n_cams)} of {world} ranks, one RCCL all_reduce of the fused BEV per step'}[shard]},
            'roofline': {'bound': 'hbm', 'kernel': timer.kernel_name, 'achieved': achieved, 'peak': HBM_PEAK_GBS,
                         'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic, 'traffic_source': traffic_src,
                         'algorithmic_bytes_per_launch': alg_bytes, 'avg_launch_us': 1e3 * avg_ms,
                         'launches_timed': len(ms),
                         'isolated': ({'avg_launch_us': 1e3 * iso_ms, 'achieved': alg_bytes / (iso_ms * 1e-3) / 1e9,
                                       'frac': alg_bytes / (iso_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                       'note': 'same kernel, stages not overlapped (after the timed region)'}
                                      if iso_ms else None)},
        }
        if cfg.render and avg_ms > 0:
            lane_slots = 256 * 4 * 16 * 2.4e9          # CUs x SIMDs x lanes x clock (MI355X_MICROARCH.md)
            out['roofline']['valu'] = {
                'pixel_records_per_launch': evals, 'pixel_records_per_sec': evals / (avg_ms * 1e-3),
                'issue_slots_per_pixel_record': 23.9,    # ISA count of the inner loop, DESIGN 4.3
                'frac_of_valu_issue_peak': evals * 23.9 / (avg_ms * 1e-3) / lane_slots,
                'frac_of_valu_issue_peak_isolated': (evals * 23.9 / (iso_ms * 1e-3) / lane_slots) if iso_ms else None}
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(hp, depth, feat, args.cpu_seconds)
            out['gpu_over_cpu'] = out['value'] / out['cpu_baseline']['value']
        print(json.dumps(out), flush=True)
    timer.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
