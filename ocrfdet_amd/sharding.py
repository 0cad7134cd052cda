"""Multi-GPU sharding of the hot path: one process per GPU, ``torch.distributed`` (backend
``nccl`` = RCCL over xGMI on MI355X; ``gloo`` in the CPU tests).

``bench.py --gpus N`` defaults to the layout of the reference's own multi-GPU runs
(tools/dist_test.sh -> MMDistributedDataParallel): every rank owns whole SAMPLES; the hot path has
no exchange step in that layout and nothing in this module is called.  The policies below split ONE
sample (or one multi-frame sequence) over ranks, which does need an exchange.

The unit of work is one camera-frame.  ``bev_pool`` is a sum over (camera, depth, pixel)
contributions per voxel (mmdet3d/ops/bev_pool_v2/src/bev_pool_cuda.cu:39-43) and frames are
independent until the channel concat (mmdet3d/models/detectors/ocrfdet.py:274), so:

* policy ``'frame'``  — a rank owns whole frames (all cameras).  No data-path collective is needed
  for the pools, the render or HOA; the per-frame BEV tensors are exchanged ONCE, by
  ``gather_frames`` (all_gather), for the consumer of the concatenated BEV.  Used whenever
  ``n_frames >= world_size``: xGMI is point-to-point (7 links x ~153 GB/s per GPU), so the cheapest
  exchange is the one that moves each finished BEV exactly once.
* policy ``'camera'`` — the cameras of a frame are split over ranks (1 camera per GPU for a 6-camera
  frame on 6 of 8 GPUs, BASELINE.json configs[3]); every rank pools only its cameras into a
  full-size partial BEV and ONE ``all_reduce(sum)`` of the fused buffer
  ``(frames, Z*C_lss + C_ht, Y, X)`` produces the BEV grids everywhere (``reduce_partial_bev``).
  fp32 summation order differs from the single-GPU order: compare at 1e-4, not bit-exact.

Nothing here computes: the functions only decide ownership and call collectives on tensors
produced by the HIP ops.
"""
import torch
import torch.distributed as dist

__all__ = ['assign_units', 'choose_policy', 'frames_of_rank', 'cams_of_rank', 'reduce_partial_bev',
           'gather_frames']


def choose_policy(n_frames, world_size):
    return 'frame' if n_frames >= world_size and n_frames % world_size == 0 else 'camera'


def assign_units(n_cams, n_frames, world_size, policy='auto'):
    """-> list over ranks of lists of (frame, cam) units.  Every unit appears exactly once."""
    if policy == 'auto':
        policy = choose_policy(n_frames, world_size)
    units = [[] for _ in range(world_size)]
    if policy == 'frame':
        for f in range(n_frames):
            units[f % world_size].extend((f, n) for n in range(n_cams))
    elif policy == 'camera':
        for f in range(n_frames):
            for n in range(n_cams):
                units[(f * n_cams + n) % world_size].append((f, n))
    else:
        raise ValueError(f'unknown policy {policy!r}')
    return units


def frames_of_rank(units):
    return sorted({f for f, _ in units})


def cams_of_rank(units, frame):
    return sorted(n for f, n in units if f == frame)


def reduce_partial_bev(partial, group=None, async_op=False):
    """The single collective of the camera policy: in-place ``all_reduce(sum)`` of this rank's
    partial fused BEV buffer (frames, channels, Y, X) (zeros for frames it holds no camera of)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return None
    return dist.all_reduce(partial, op=dist.ReduceOp.SUM, group=group, async_op=async_op)


def gather_frames(local_frames, n_frames, group=None):
    """The single collective of the frame policy: every rank contributes its finished per-frame
    BEV tensors (n_local, channels, Y, X) (frame f lives on rank f % world) and receives all
    ``n_frames`` of them in frame order, i.e. the operand of the reference's channel concat."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local_frames
    world = dist.get_world_size(group)
    n_local = n_frames // world
    assert local_frames.shape[0] == n_local, 'frame policy needs n_frames % world_size == 0'
    out = torch.empty((world * n_local,) + tuple(local_frames.shape[1:]), dtype=local_frames.dtype,
                      device=local_frames.device)
    dist.all_gather_into_tensor(out, local_frames.contiguous(), group=group)
    # out[r*n_local + j] is frame j*world + r  ->  frame-major order
    out = out.view((world, n_local) + tuple(local_frames.shape[1:]))
    return out.transpose(0, 1).reshape((n_frames,) + tuple(local_frames.shape[1:]))
