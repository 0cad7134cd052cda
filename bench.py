#!/usr/bin/env python3
"""bench.py — throughput of the OcRF render + BEV-pool + HOA hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config NAME]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one pass of the hot path (ocrfdet_amd.hotpath.HotPath.step) over one batch of
synthetic input (seeded) that is already resident in HBM.  Rank 0 prints ONE JSON
line; `value` = BEV voxels written per second by the whole job (both pools executed per voxel
grid), with `roofline` for the dominant kernel (HIP-event timed inside the timed region) and
`cpu_baseline` (the C oracle on this box's host cores, rank 0, N=1 only).
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
DEFAULT_CONFIG = 'cfg1_6cam_256x704_bev128x128x8'


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--config', default=DEFAULT_CONFIG)
    ap.add_argument('--cpu-seconds', type=float, default=12.0, help='CPU-baseline sample budget')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    return ap.parse_args()


def cpu_baseline(hp, depth, feat, budget_s):
    """Times the C oracle (oracle/bev_pool_ref.c, OpenMP) on the same step: LSS pool + HT pool."""
    import numpy as np
    import oracle
    oracle.build()
    d, f = depth.cpu().numpy(), feat.cpu().numpy()
    plans = []
    for p in (hp.lss, hp.ht):
        plans.append(tuple(t.cpu().numpy() for t in (p.ranks_depth, p.ranks_feat, p.ranks_bev, p.starts, p.lengths)) + (p.bev_shape,))

    def one():
        for rd, rf, rb, st, ln, shape in plans:
            oracle.bev_pool_v2(d, f, rd, rf, rb, shape, st, ln)     # includes the wrapper's permute
    one()                                                            # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        one()
        n += 1
        el = time.perf_counter() - t0
        if el >= budget_s or n >= 400:
            break
    return dict(value=hp.bev_voxels_per_step * n / el, unit='BEV voxels/s', cores=oracle.num_threads(),
                kind='port', ms_per_step=1e3 * el / n,
                sample=f'{n} full steps (LSS pool + HT pool, same inputs and ranks as the GPU step) '
                       f'in {el:.1f} s with the C/OpenMP oracle')


def main():
    args = parse()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)

    from ocrfdet_amd import _lib, hotpath, synthetic
    _lib.lib()                      # raises if libocrf_hip.so is missing — no fallback
    cfg = synthetic.CONFIGS[args.config]
    # weak scaling: every rank owns cfg.n_frames frames of a (world * n_frames)-frame sequence;
    # frames are independent until the channel concat (detectors/ocrfdet.py:274)
    hp = hotpath.HotPath(cfg, dev)
    depth, feat = hp.make_inputs(seed=rank)
    gathered = None
    if world > 1:
        lss0, ht0 = hp.step(depth, feat)
        both = torch.cat((lss0, ht0), 1)
        gathered = torch.empty((world,) + tuple(both.shape), device=dev)

    def step():
        lss, ht = hp.step(depth, feat)
        if world > 1:
            dist.all_gather_into_tensor(gathered, torch.cat((lss, ht), 1))
        return lss, ht

    for _ in range(args.warmup):
        step()
    timer = _lib.KernelTimer(_lib.K_BEV_POOL_FWD, 2 * args.steps)
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

    if rank == 0:
        ms = timer.read_ms()
        avg_ms = sum(ms) / max(len(ms), 1)
        alg_bytes = 0.5 * (hp.lss.algorithmic_bytes(depth.numel(), feat.numel()) +
                           hp.ht.algorithmic_bytes(depth.numel(), feat.numel()))
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        voxels = hp.bev_voxels_per_step * world * args.steps
        out = {
            'metric': 'BEV voxels/sec + rendered views/sec, 6-cam 256x704',
            'value': voxels / elapsed, 'unit': 'BEV voxels/s',
            'rendered_views_per_sec': 0.0,
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': 1e3 * elapsed / args.steps, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': cfg.name, 'cams': cfg.n_cams, 'frames_per_gpu': cfg.n_frames,
                       'bev': list(cfg.bev_xyz), 'channels': cfg.channels, 'depth_bins': cfg.D,
                       'stages': 'lss_pool+ht_pool', 'index_prep': 'cached (accelerate=True semantics)',
                       'sharding': f'frame-shard x{world}, all_gather of per-frame BEV' if world > 1 else 'none'},
            'roofline': {'bound': 'hbm', 'kernel': timer.kernel_name, 'achieved': achieved, 'peak': HBM_PEAK_GBS,
                         'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS, 'traffic': None,
                         'algorithmic_bytes_per_launch': alg_bytes, 'avg_launch_us': 1e3 * avg_ms,
                         'launches_timed': len(ms)},
        }
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(hp, depth, feat, args.cpu_seconds)
            out['gpu_over_cpu'] = out['value'] / out['cpu_baseline']['value']
        print(json.dumps(out), flush=True)
    timer.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
