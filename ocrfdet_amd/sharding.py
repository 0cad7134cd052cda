"""Multi-GPU sharding of the hot path: one process per GPU, ``torch.distributed`` (backend
``nccl`` = RCCL over xGMI on MI355X; ``gloo`` in the CPU tests).

The unit of work is one CAMERA-FRAME.  ``bev_pool`` is a sum over (camera, depth, pixel) contributions per
voxel (mmdet3d/ops/bev_pool_v2/src/bev_pool_cuda.cu:39-43), so the cameras of a frame may be pooled on
different GPUs into full-size partial BEVs that are then summed; frames are independent until the channel
concat (mmdet3d/models/detectors/ocrfdet.py:274), so different frames never need a reduction.

``CameraFramePlan`` (``bench.py --gpus N`` default, BASELINE.json north_star): the ``n_frames x n_cams``
camera-frames of ONE sample over ALL ``world`` ranks.
* ``world <= n_frames``: a rank owns whole frames — no reduction at all;
* ``world  > n_frames``: the ranks form one group of ``G = world / n_frames`` per frame and the frame's
  cameras are dealt round-robin over its group (6 cameras over 4 ranks: 2, 2, 1, 1).
The exchange into the fused BEV grid ``(n_frames, P = Z*C + C, Y, X)`` (LSS planes, then HT planes) is
  1. ``reduce_scatter`` INSIDE a frame's group over plane blocks (member i receives planes
     ``[i P/G, (i+1) P/G)`` of that frame summed over the group's cameras) — only for G > 1;
  2. ONE ``all_gather`` over the world of every rank's finished plane block, which leaves the whole fused
     grid on every rank (the consumer — HOA gate, BEV encoder — may run replicated or on one rank).
xGMI is point-to-point (7 links per GPU): both steps move ``P Y X 4 / G`` resp. ``/ world`` bytes per link
with all links busy at once, instead of the ``2 (N-1)/N`` full-buffer ring of an ``all_reduce``; both are
issued asynchronously (RCCL's own stream) so that the renders and HOA-1/2, which do not depend on the pooled
BEV, run beside them.  fp32 summation order differs from the single-GPU order: compare at 1e-4, not bit-exact.

The older policies (``assign_units`` / ``reduce_partial_bev`` / ``gather_frames``) are kept: ``'frame'``
for multi-frame SEQUENCES with one all_gather, ``'camera'`` with one dense all_reduce.

Nothing here computes: the functions only decide ownership and call collectives on tensors produced by the
HIP ops.
"""
import torch
import torch.distributed as dist

__all__ = ['assign_units', 'choose_policy', 'frames_of_rank', 'cams_of_rank', 'reduce_partial_bev',
           'gather_frames', 'CameraFramePlan', 'BevExchange']


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


class CameraFramePlan:
    """Ownership of the camera-frames of one sample over ``world`` ranks (see the module docstring).

    ``units[r]``          list of (frame, cam) owned by rank r (every camera-frame exactly once);
    ``group_of_frame[f]`` the ranks that hold cameras of frame f, in block order;
    ``block_owner``       list of (frame, first plane, n planes, rank): who holds which finished plane block of
                          the fused grid after step 1 — the layout of the world all_gather."""

    def __init__(self, n_cams, n_frames, world, n_planes):
        if world < 1 or n_cams < 1 or n_frames < 1:
            raise ValueError('world, n_cams and n_frames must be positive')
        self.n_cams, self.n_frames, self.world, self.n_planes = n_cams, n_frames, world, n_planes
        self.units = [[] for _ in range(world)]
        self.group_of_frame = []
        if world <= n_frames:
            for f in range(n_frames):
                r = f % world
                self.units[r].extend((f, c) for c in range(n_cams))
                self.group_of_frame.append([r])
        else:
            # G ranks per frame; a remainder (world not a multiple of n_frames) goes to the first frames
            base, extra = divmod(world, n_frames)
            start = 0
            for f in range(n_frames):
                G = min(base + (1 if f < extra else 0), n_cams)
                ranks = list(range(start, start + G))
                start += base + (1 if f < extra else 0)
                for c in range(n_cams):
                    self.units[ranks[c % G]].append((f, c))
                self.group_of_frame.append(ranks)
        # plane blocks: frame f's P planes in G_f blocks (sizes differ by at most one when G_f does not divide P)
        self.block_owner = []
        for f, ranks in enumerate(self.group_of_frame):
            G = len(ranks)
            q, rem = divmod(n_planes, G)
            p0 = 0
            for i, r in enumerate(ranks):
                n = q + (1 if i < rem else 0)
                self.block_owner.append((f, p0, n, r))
                p0 += n

    def frames_of(self, rank):
        return sorted({f for f, _ in self.units[rank]})

    def cams_of(self, rank, frame):
        return sorted(c for f, c in self.units[rank] if f == frame)

    def blocks_of(self, rank):
        return [(f, p0, n) for f, p0, n, r in self.block_owner if r == rank]

    @property
    def idle_ranks(self):
        return [r for r in range(self.world) if not self.units[r]]

    def describe(self):
        g = [len(x) for x in self.group_of_frame]
        return (f'{self.n_frames} frame(s) x {self.n_cams} cameras over {self.world} ranks: '
                f'{"whole frames per rank, no reduction" if max(g) == 1 else f"groups of {g} ranks per frame"}; '
                f'camera-frames per rank {[len(u) for u in self.units]}')


class BevExchange:
    """The collectives of a ``CameraFramePlan`` on this rank.

        ex = BevExchange(plan, rank, device, (Y, X))
        for f in plan.frames_of(rank): pool the owned cameras of frame f into ex.pool_target(f)   # (P, Y, X)
        works = ex.start()            # reduce_scatter inside each frame group with > 1 rank (asynchronous)
        ... kernels that do not need the pooled BEV ...
        full = ex.finish(works)       # world all_gather -> (n_frames, P, Y, X), the same on every rank

    ``pool_target(f)`` is a full-size partial buffer when the frame's cameras are split over a group, and the
    frame's final place inside ``full`` when this rank owns the whole frame (nothing is copied then).  Where the
    plane blocks lie in rank order (world <= n_frames == world, or world a multiple of n_frames with P divisible
    by the group size) the all_gather runs IN PLACE on ``full``; otherwise through a staging slot per rank.
    Process groups are created collectively by every rank in the same order (constructor).  ``gloo`` has no
    reduce_scatter and no in-place gather: there step 1 is an all_reduce of the frame buffer followed by taking
    the own block, and the gather is staged (CPU tests only)."""

    def __init__(self, plan, rank, device, plane_shape, dtype=torch.float32):
        self.plan, self.rank, self.device = plan, rank, torch.device(device)
        self.Y, self.X = plane_shape
        self.active = dist.is_initialized() and plan.world > 1
        self.backend = dist.get_backend() if self.active else None
        self.groups = {}
        if self.active:
            for ranks in plan.group_of_frame:
                key = tuple(ranks)
                if len(ranks) > 1 and key not in self.groups:
                    self.groups[key] = dist.new_group(ranks=ranks)      # every rank calls this, members or not
        P, W = plan.n_planes, plan.world
        kw = dict(dtype=dtype, device=self.device)
        self.my_blocks = plan.blocks_of(rank)
        self.slot = max(sum(n for _, _, n in plan.blocks_of(r)) for r in range(W))
        self.full = torch.empty(plan.n_frames, P, self.Y, self.X, **kw)
        # blocks in rank order == planes of `full` in memory order?
        flat, ok = 0, (W * self.slot == plan.n_frames * P)
        for r in range(W):
            for f, p0, n in plan.blocks_of(r):
                ok = ok and (f * P + p0 == flat)
                flat += n
        self.direct = ok and self.backend != 'gloo'
        if self.direct:
            self.recv = self.full.view(W * self.slot, self.Y, self.X)
            self.send = self.recv[rank * self.slot:(rank + 1) * self.slot]
        else:
            self.send = torch.zeros(self.slot, self.Y, self.X, **kw)
            self.recv = torch.empty(W * self.slot, self.Y, self.X, **kw)
        if self.direct and self.active and not self._in_place_gather_works():
            self.direct = False                  # staged gather: one extra copy of the grid, always valid
            self.send = torch.zeros(self.slot, self.Y, self.X, **kw)
            self.recv = torch.empty(W * self.slot, self.Y, self.X, **kw)
        self.partial = {f: torch.empty(P, self.Y, self.X, **kw) for f in plan.frames_of(rank)
                        if len(plan.group_of_frame[f]) > 1}
        esz = self.full.element_size()
        self.bytes_reduce_scatter = sum((len(plan.group_of_frame[f]) - 1) * (P // len(plan.group_of_frame[f])) *
                                        self.Y * self.X * esz for f in self.partial)
        self.bytes_all_gather = (W - 1) * self.slot * self.Y * self.X * esz if W > 1 else 0

    def _in_place_gather_works(self):
        """One trial of the in-place all_gather (send block = this rank's slice of the receive buffer, the form NCCL /
        RCCL document as in-place) with known values, at construction; every rank reaches the same verdict."""
        try:
            self.send.fill_(float(self.rank + 1))
            dist.all_gather_into_tensor(self.recv, self.send)
            got = self.recv.view(self.plan.world, -1)[:, 0].float().cpu()
            ok = bool(torch.equal(got, torch.arange(1, self.plan.world + 1, dtype=torch.float32)))
        except Exception:
            ok = False
        flag = torch.tensor([1 if ok else 0], device=self.device, dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return bool(flag.item())

    def _send_block(self, f):
        off = 0
        for bf, p0, n in self.my_blocks:
            if bf == f:
                return self.send[off:off + n], p0, n
            off += n
        raise KeyError(f)

    def pool_target(self, f):
        """(P, Y, X) tensor the pools of frame ``f`` on this rank write (LSS planes first, then HT)."""
        if f in self.partial:
            return self.partial[f]
        dst, p0, n = self._send_block(f)
        assert p0 == 0 and n == self.plan.n_planes
        return dst

    def start(self):
        """Step 1 for every frame whose cameras this rank shares with others.  -> outstanding works."""
        works = []
        for f, buf in self.partial.items():
            ranks = self.plan.group_of_frame[f]
            dst, p0, n = self._send_block(f)
            if not self.active:
                dst.copy_(buf[p0:p0 + n])
                continue
            group, G = self.groups[tuple(ranks)], len(ranks)
            if self.backend == 'gloo' or self.plan.n_planes % G != 0:
                works.append((dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group, async_op=True), dst, buf[p0:p0 + n]))
            else:
                works.append((dist.reduce_scatter_tensor(dst, buf, op=dist.ReduceOp.SUM, group=group, async_op=True),
                              None, None))
        return works

    def finish(self, works=()):
        """Waits for step 1 (on the current stream), then the world all_gather; -> (n_frames, P, Y, X)."""
        for w, dst, src in works:
            w.wait()
            if dst is not None:
                dst.copy_(src)
        if self.active:
            dist.all_gather_into_tensor(self.recv, self.send)
        if not self.direct:
            slots = self.recv.view(self.plan.world, self.slot, self.Y, self.X) if self.active else self.send.unsqueeze(0)
            off = [0] * self.plan.world
            for f, p0, n, r in self.plan.block_owner:
                if n:
                    self.full[f, p0:p0 + n].copy_(slots[r if self.active else 0, off[r]:off[r] + n])
                off[r] += n
        return self.full
