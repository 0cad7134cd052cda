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
           'gather_frames', 'CameraFramePlan', 'BevExchange', 'PipelinedExchange', 'gate_blocks', 'Collectives', 'GlooCollectives',
           'GlooEmulation']


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
        # plane blocks: frame f's P planes in G_f blocks of cap_f = ceil(P / G_f) planes (the last one shorter when G_f
        # does not divide P): equal-size blocks are what reduce_scatter moves, so a group of 3 (6 ranks x 2 frames,
        # BASELINE configs[3]: P = 160) keeps the scatter form with two padding planes instead of a dense all_reduce
        self.block_owner = []
        self.cap_of_frame = []
        for f, ranks in enumerate(self.group_of_frame):
            G = len(ranks)
            cap = -(-n_planes // G)
            self.cap_of_frame.append(cap)
            for i, r in enumerate(ranks):
                p0 = min(i * cap, n_planes)
                self.block_owner.append((f, p0, min(cap, n_planes - p0), r))

    def frames_of(self, rank):
        return sorted({f for f, _ in self.units[rank]})

    def cams_of(self, rank, frame):
        return sorted(c for f, c in self.units[rank] if f == frame)

    def blocks_of(self, rank):
        return [(f, p0, n) for f, p0, n, r in self.block_owner if r == rank]

    @property
    def any_shared(self):
        """True when some frame's cameras are split over several ranks (a function of the plan: the same on every rank)."""
        return any(len(g) > 1 for g in self.group_of_frame)

    @property
    def idle_ranks(self):
        return [r for r in range(self.world) if not self.units[r]]

    def describe(self):
        g = [len(x) for x in self.group_of_frame]
        return (f'{self.n_frames} frame(s) x {self.n_cams} cameras over {self.world} ranks: '
                f'{"whole frames per rank, no reduction" if max(g) == 1 else f"groups of {g} ranks per frame"}; '
                f'camera-frames per rank {[len(u) for u in self.units]}')


class Collectives:
    """The ``torch.distributed`` calls ``BevExchange`` issues, behind one seam: the CPU tests drive the exchange's RCCL
    code paths (in-place gather, ``reduce_scatter_tensor``) over gloo through ``GlooEmulation``."""
    native = True            # has reduce_scatter_tensor and the in-place all_gather (nccl = RCCL)

    def reduce_scatter(self, dst, src, group):
        return dist.reduce_scatter_tensor(dst, src, op=dist.ReduceOp.SUM, group=group, async_op=True)

    def all_gather(self, recv, send, in_place):
        dist.all_gather_into_tensor(recv, send)

    def exchange(self, ops):
        return dist.batch_isend_irecv(ops) if ops else []


class GlooCollectives(Collectives):
    """gloo has no reduce_scatter and no in-place gather: step 1 is an all_reduce of the (padded) frame buffer followed
    by taking the own block, and the gather is staged.  CPU plumbing only."""
    native = False

    def reduce_scatter(self, dst, src, group):
        dist.all_reduce(src, op=dist.ReduceOp.SUM, group=group)
        n = dst.shape[0]
        i = dist.get_rank(group)
        dst.copy_(src[i * n:(i + 1) * n])
        return None


class GlooEmulation(GlooCollectives):
    """Test seam: reports itself as native so that ``BevExchange`` takes exactly the branches it takes on RCCL (views of
    the fused grid as gather slots, ``reduce_scatter`` into them), with the two collectives gloo lacks emulated."""
    native = True

    def all_gather(self, recv, send, in_place):
        dist.all_gather_into_tensor(recv, send.clone() if in_place else send)


class BevExchange:
    """The collectives of a ``CameraFramePlan`` on this rank.

        ex = BevExchange(plan, rank, device, (Y, X))
        for f in plan.frames_of(rank): pool the owned cameras of frame f into ex.pool_target(f)   # (P, Y, X)
        works = ex.start()            # step 1 inside each frame group with > 1 rank (asynchronous)
        ... kernels that do not need the pooled BEV ...
        full = ex.finish(works)       # world all_gather -> (n_frames, P, Y, X), the same on every rank

    ``pool_target(f)`` is a full-size partial buffer when the frame's cameras are split over a group, and the
    frame's final place inside ``full`` when this rank owns the whole frame (nothing is copied then).
    Step 1, dense: ``reduce_scatter`` over equal plane blocks of ``cap = ceil(P / G)`` planes (the partial buffer carries
    ``G cap - P`` zero padding planes, so any group size keeps the scatter form).  Step 1, wedge-sparse
    (``set_touched``): a camera's frustum covers a wedge of the BEV, so a member's partial grid is zero outside the
    strips its cameras' rank vectors touch (static per calibration); it sends every other member only the touched
    strips of that member's plane block (one batched isend / irecv round: all links at once), and the receiver adds
    the contributions in member order (deterministic).
    Step 2: where the plane blocks lie in rank order (world == n_frames, or world a multiple of n_frames with P
    divisible by the group size) the all_gather runs IN PLACE on ``full``; otherwise through a staging slot per rank.
    Process groups are created collectively by every rank in the same order (constructor)."""

    def __init__(self, plan, rank, device, plane_shape, dtype=torch.float32, collectives=None, share=None):
        """``share``: another ``BevExchange`` of the same plan whose process groups (and in-place-gather verdict) this one
        reuses instead of creating its own — a second set of BUFFERS on the same communicators (``PipelinedExchange``)."""
        self.plan, self.rank, self.device = plan, rank, torch.device(device)
        self.Y, self.X = plane_shape
        self.active = dist.is_initialized() and plan.world > 1
        self.backend = dist.get_backend() if self.active else None
        if collectives is None:
            collectives = share.c if share is not None else (GlooCollectives() if self.backend == 'gloo' else Collectives())
        self.c = collectives
        self.groups = {}
        self.ctrl = None
        if share is not None:
            self.groups, self.ctrl = share.groups, share.ctrl
        elif self.active:
            # control plane: a gloo group of its own for host-side agreement (flags, index lists), so that a verdict about
            # a data-path collective never travels over the communicator that collective may just have failed on
            self.ctrl = dist.new_group(backend='gloo')
            for ranks in plan.group_of_frame:
                key = tuple(ranks)
                if len(ranks) > 1 and key not in self.groups:
                    self.groups[key] = dist.new_group(ranks=ranks)      # every rank calls this, members or not
        P, W = plan.n_planes, plan.world
        kw = dict(dtype=dtype, device=self.device)
        # blocks of a rank sit in its gather slot at offsets of their CAPACITY (cap planes each; n <= cap are real)
        self.my_blocks = plan.blocks_of(rank)
        cap_sum = lambda r: sum(plan.cap_of_frame[f] if len(plan.group_of_frame[f]) > 1 else n  # noqa: E731
                                for f, _, n in plan.blocks_of(r))
        self.slot = max(cap_sum(r) for r in range(W))
        self.full = torch.empty(plan.n_frames, P, self.Y, self.X, **kw)
        # blocks in rank order == planes of `full` in memory order (and no padding anywhere)?
        flat, ok = 0, (W * self.slot == plan.n_frames * P)
        for r in range(W):
            for f, p0, n in plan.blocks_of(r):
                ok = ok and (f * P + p0 == flat)
                flat += n
        self.direct = ok and self.c.native
        if self.direct:
            self.recv = self.full.view(W * self.slot, self.Y, self.X)
            self.send = self.recv[rank * self.slot:(rank + 1) * self.slot]
        else:
            self.send = torch.zeros(self.slot, self.Y, self.X, **kw)
            self.recv = torch.empty(W * self.slot, self.Y, self.X, **kw)
        if self.direct and self.active and not (share.direct if share is not None else self._in_place_gather_works()):
            self.direct = False                  # staged gather: one extra copy of the grid, always valid
            self.send = torch.zeros(self.slot, self.Y, self.X, **kw)
            self.recv = torch.empty(W * self.slot, self.Y, self.X, **kw)
        # partial buffers of shared frames: G * cap planes, the padding planes stay zero (the pools write [:P])
        self.partial = {f: torch.zeros(len(plan.group_of_frame[f]) * plan.cap_of_frame[f], self.Y, self.X, **kw)
                        for f in plan.frames_of(rank) if len(plan.group_of_frame[f]) > 1}
        self.touched = {}                        # frame -> list over the group's members of tile index tensors
        self._flat = {}                          # frame -> list over the members of the tiles' flat voxel offsets
        # sparsity granule: 8 x 8 voxel blocks of the (Y, X) plane (a camera wedge touches ~30 % of them), rows otherwise
        self.tile_side = 8 if self.Y % 8 == 0 and self.X % 8 == 0 else 0
        self.strip = 64 if self.tile_side else self.X            # voxels per tile
        self.n_strips = self.Y * self.X // self.strip            # tiles per plane
        self._account()

    def tile_of_voxel(self, yx):
        """Tile index of flattened (Y, X) voxel offsets (int64 tensor)."""
        if self.tile_side:
            y, x = yx // self.X, yx % self.X
            return (y // 8) * (self.X // 8) + x // 8
        return yx // self.X

    def _voxels_of_tiles(self, tiles):
        """(len(tiles) * strip,) flat voxel offsets of the given tiles, tile-major."""
        t = tiles.long()
        if self.tile_side:
            ty, tx = t // (self.X // 8), t % (self.X // 8)
            d = torch.arange(8, device=t.device)
            yy = (ty[:, None, None] * 8 + d[None, :, None]) * self.X
            xx = tx[:, None, None] * 8 + d[None, None, :]
            return (yy + xx).reshape(-1)
        return (t[:, None] * self.X + torch.arange(self.X, device=t.device)[None, :]).reshape(-1)

    def _account(self):
        plan, esz = self.plan, self.full.element_size()
        W = plan.world
        rs = 0
        for f in self.partial:
            G, cap = len(plan.group_of_frame[f]), plan.cap_of_frame[f]
            if f in self.touched:
                _, _, n_me = self._my_block(f)
                me = plan.group_of_frame[f].index(self.rank)
                rs += sum(int(t.numel()) for j, t in enumerate(self.touched[f]) if j != me) * self.strip * n_me * esz
            else:
                rs += (G - 1) * cap * self.Y * self.X * esz
        self.bytes_reduce_scatter = rs
        self.bytes_reduce_scatter_dense = sum((len(plan.group_of_frame[f]) - 1) * plan.cap_of_frame[f] * self.Y * self.X *
                                              esz for f in self.partial)
        self.bytes_all_gather = (W - 1) * self.slot * self.Y * self.X * esz if W > 1 else 0

    def _in_place_gather_works(self):
        """One trial of the in-place all_gather (send block = this rank's slice of the receive buffer, the form NCCL /
        RCCL document as in-place) with known values, at construction; every rank reaches the same verdict."""
        try:
            self.send.fill_(float(self.rank + 1))
            self.c.all_gather(self.recv, self.send, True)
            got = self.recv.view(self.plan.world, -1)[:, 0].float().cpu()
            ok = bool(torch.equal(got, torch.arange(1, self.plan.world + 1, dtype=torch.float32)))
        except Exception:
            ok = False
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32)              # host tensor, gloo control group
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.ctrl)
        return bool(flag.item())

    def _my_block(self, f):
        """-> (slot offset, first plane, real planes) of this rank's block of frame f."""
        off = 0
        for bf, p0, n in self.my_blocks:
            shared = len(self.plan.group_of_frame[bf]) > 1
            if bf == f:
                return off, p0, n
            off += self.plan.cap_of_frame[bf] if shared else n
        raise KeyError(f)

    def pool_target(self, f):
        """(P, Y, X) tensor the pools of frame ``f`` on this rank write (LSS planes first, then HT)."""
        P = self.plan.n_planes
        if f in self.partial:
            return self.partial[f][:P]
        off, p0, n = self._my_block(f)
        assert p0 == 0 and n == P
        return self.send[off:off + n]

    def adopt_touched(self, other):
        """Take over another exchange's (same plan, same rank) touched-tile lists without a second collective."""
        self.touched = {f: [t.clone() for t in ts] for f, ts in other.touched.items()}
        self._flat = {f: [self._voxels_of_tiles(t) for t in ts] for f, ts in self.touched.items()}
        self._account()

    def set_touched(self, touched):
        """Switch step 1 to the wedge-sparse form.  ``touched``: {frame: 1-D int64 tensor of the tile indices
        (``tile_of_voxel``: 8 x 8 voxel blocks of the (Y, X) plane, or rows when Y or X is no multiple of 8; ``ex.strip``
        voxels each, ``ex.n_strips`` per plane) this rank's cameras of that frame can write} for every shared frame this
        rank holds — an EMPTY dict on a rank that shares no frame (idle ranks, owners of whole frames).  Collective over
        the world on the gloo control group: EVERY rank of the job calls it (the index lists of a group's members are
        exchanged once: they are static per calibration)."""
        mine = {int(f): torch.as_tensor(t).long().cpu().unique().tolist() for f, t in touched.items() if f in self.partial}
        if self.active:
            everyone = [None] * self.plan.world
            dist.all_gather_object(everyone, mine, group=self.ctrl)
        else:
            everyone = [mine]
        self.touched = {}
        for f in self.partial:
            ranks = self.plan.group_of_frame[f]
            lists = [everyone[r if self.active else 0].get(f) for r in ranks] if self.active else [mine.get(f)]
            if any(x is None for x in lists):
                raise ValueError(f'set_touched: a member of frame {f}\'s group gave no strip list')
            self.touched[f] = [torch.tensor(x, dtype=torch.long, device=self.device) for x in lists]
            self._flat[f] = [self._voxels_of_tiles(t) for t in self.touched[f]]
        self._account()

    def start(self):
        """Step 1 for every frame whose cameras this rank shares with others.  -> outstanding works."""
        works = []
        for f, buf in self.partial.items():
            ranks = self.plan.group_of_frame[f]
            off, p0, n = self._my_block(f)
            cap = self.plan.cap_of_frame[f]
            dst = self.send[off:off + cap]
            if not self.active:
                dst[:n].copy_(buf[p0:p0 + n])
                continue
            group = self.groups[tuple(ranks)]
            if f in self.touched:
                works.append(self._start_sparse(f, buf, dst, ranks, group, cap))
            else:
                works.append(('dense', self.c.reduce_scatter(dst, buf, group)))
        return works

    def _start_sparse(self, f, buf, dst, ranks, group, cap):
        me = ranks.index(self.rank)
        L = self.strip
        idx_me, flat_me = self.touched[f][me], self._flat[f][me]
        P = self.plan.n_planes
        ops, recv_bufs, keep = [], {}, []
        for j, r in enumerate(ranks):
            p0j = min(j * cap, P)
            nj = min(cap, P - p0j)
            if j == me or nj == 0 or idx_me.numel() == 0:
                continue
            chunk = buf[p0j:p0j + nj].view(nj, -1).index_select(1, flat_me)                 # touched tiles of j's block
            keep.append(chunk)
            ops.append(dist.P2POp(dist.isend, chunk, r, group))
        p0 = min(me * cap, P)
        n_me = min(cap, P - p0)
        for j, r in enumerate(ranks):
            tj = self.touched[f][j]
            if j == me or n_me == 0 or tj.numel() == 0:
                continue
            recv_bufs[j] = torch.empty(n_me, int(tj.numel()) * L, dtype=buf.dtype, device=buf.device)
            ops.append(dist.P2POp(dist.irecv, recv_bufs[j], r, group))
        reqs = self.c.exchange(ops)
        return ('sparse', reqs, f, buf, dst, me, p0, n_me, recv_bufs, keep)

    def _finish_sparse(self, work):
        _, reqs, f, buf, dst, me, p0, n_me, recv_bufs, _keep = work
        for q in reqs:
            q.wait()
        if n_me == 0:
            return
        out = dst[:n_me].view(n_me, -1)
        out.zero_()
        for j, tj in enumerate(self.touched[f]):                 # member order: the sum is deterministic
            if tj.numel() == 0:
                continue
            fj = self._flat[f][j]
            src = buf[p0:p0 + n_me].view(n_me, -1).index_select(1, fj) if j == me else recv_bufs[j]
            out.index_add_(1, fj, src)                           # a member's offsets are unique: no accumulation order inside

    def finish(self, works=()):
        """Waits for step 1 (on the current stream), then the world all_gather; -> (n_frames, P, Y, X)."""
        self.finish_reduce(works)
        return self.gather()

    def finish_reduce(self, works=()):
        """Waits for step 1 on the current stream: this rank's plane blocks (``block_views``) hold their final sums."""
        for w in works:
            if w[0] == 'sparse':
                self._finish_sparse(w)
            elif w[1] is not None:
                w[1].wait()

    def block_views(self):
        """-> [(frame, first plane, n planes, (n, Y, X) view)]: this rank's finished plane blocks where the world
        all_gather will pick them up — what is written there (a gate applied in place, an extra plane filled in) travels."""
        out = []
        for f, p0, n in self.my_blocks:
            off, _, _ = self._my_block(f)
            out.append((f, p0, n, self.send[off:off + n]))
        return out

    def group_gather(self, f, t):
        """all_gather of a small tensor inside frame f's group -> (G, *t.shape), members in block order (G = 1, or no
        process group: the tensor alone)."""
        ranks = self.plan.group_of_frame[f]
        if not self.active or len(ranks) == 1:
            return t.unsqueeze(0)
        out = torch.empty((len(ranks),) + tuple(t.shape), dtype=t.dtype, device=t.device)
        dist.all_gather_into_tensor(out.view(-1), t.contiguous().view(-1), group=self.groups[tuple(ranks)])      # (flat: both backends)
        return out

    def gather(self):
        """Step 2: the world all_gather of every rank's plane blocks; -> (n_frames, P, Y, X), the same on every rank."""
        if self.active:
            self.c.all_gather(self.recv, self.send, self.direct)
        if not self.direct:
            slots = self.recv.view(self.plan.world, self.slot, self.Y, self.X) if self.active else self.send.unsqueeze(0)
            off = [0] * self.plan.world
            for f, p0, n, r in self.plan.block_owner:
                if n:
                    self.full[f, p0:p0 + n].copy_(slots[r if self.active else 0, off[r]:off[r] + n])
                off[r] += self.plan.cap_of_frame[f] if len(self.plan.group_of_frame[f]) > 1 else n
        return self.full


def gate_blocks(ex, first_plane, n_channels, opacity_plane, opacity_of_frame, stats_fn, gate_fn, partial_ok=False):
    """HOA-3 on a camera-frame-sharded grid, BETWEEN the two steps of the exchange (``ex.finish_reduce`` done,
    ``ex.gather`` to come): every rank gates the channels it holds, in place, so that the world all_gather carries the
    GATED planes — no rank runs the gate for a frame it has no part in, and inside a frame's group the gate's work is
    split by plane block instead of replicated.

    The gate of a frame (view_transformer_ocrf.py:230-242,1197-1199) is mask = sigmoid(conv7x7([mean_c x, max_c x]) +
    opacity_bev), x * mask: the per-pixel statistics run over ALL ``n_channels`` channels of the frame's gated planes
    ``[first_plane, first_plane + n_channels)``, which a group's members hold in blocks.  So: each member reduces its
    own channels — ``stats_fn(x (c,Y,X)) -> (2,Y,X) = [mean, max] over its c channels`` —, the group all_gathers these
    (2 x Y x X floats per member: 320 KB at 200 x 200) and combines them in member order with the weights c_j / C the
    plan fixes (deterministic, the same on every member; a member that holds ALL channels gets its own statistics back
    bit for bit), then gates its channels with the combined statistics — ``gate_fn(x, stats (2,Y,X), opacity (Y,X))``,
    in place.  The member whose block holds ``opacity_plane`` (an extra plane of the fused grid, or None) stores the
    frame's opacity BEV there: it travels with the gather too.  ``opacity_of_frame``: {frame: (Y,X) tensor} for the
    frames this rank has blocks of.  Frames are independent (mmdet3d/models/detectors/ocrfdet.py:274)."""
    plan = ex.plan
    P = plan.n_planes
    for f, p0, n, view in ex.block_views():
        c0, c1 = max(p0, first_plane), min(p0 + n, first_plane + n_channels)
        x = view[c0 - p0:c1 - p0] if c1 > c0 else view[:0]
        if x.shape[0] > 0:
            part = stats_fn(x)
        else:
            part = torch.stack((torch.zeros(ex.Y, ex.X, dtype=view.dtype, device=view.device),
                                torch.full((ex.Y, ex.X), float('-inf'), dtype=view.dtype, device=view.device)))
        ranks, cap = plan.group_of_frame[f], plan.cap_of_frame[f]
        if (len(ranks) == 1 or not ex.active) and x.shape[0] == n_channels:
            # the member holds every channel of the frame: its own statistics ARE the frame's (no gather, no combination:
            # three small launches less per block)
            gate_fn(x, part, opacity_of_frame[f])
            if opacity_plane is not None and p0 <= opacity_plane < p0 + n:
                view[opacity_plane - p0].copy_(opacity_of_frame[f])
            continue
        parts = ex.group_gather(f, part)
        # channels every member of the group holds: a function of the plan (equal blocks of cap planes)
        held = []
        for j in range(len(ranks)):
            q0 = min(j * cap, P) if len(ranks) > 1 else 0
            qn = min(cap, P - q0) if len(ranks) > 1 else P
            held.append(max(0, min(q0 + qn, first_plane + n_channels) - max(q0, first_plane)))
        if parts.shape[0] != len(held):
            # the gather returned only this rank's part (an inactive exchange: no process group) although the frame's
            # channels are spread over a group: gating with the own channels' statistics alone is a wrong mask for the
            # frame — refused unless the caller says that is what it wants (``partial_ok``: the
            # single-process tests that look at ONE rank of a sharded job).  A rank that holds every channel took the
            # branch above; one that holds none has nothing to gate
            if x.shape[0] > 0 and not partial_ok:
                from . import _lib
                raise _lib.OcrfHipError(
                    f'gate_blocks: frame {f} is held by {len(ranks)} ranks but only {parts.shape[0]} part(s) of its channel '
                    f'statistics arrived and this rank holds {x.shape[0]} of {n_channels} channels: the exchange is not active')
            held = [x.shape[0]]
        mean, smax = None, None
        for j, cj in enumerate(held):                           # member order: every member forms the same sums
            if cj == 0:
                continue
            term = parts[j, 0] * (float(cj) / float(n_channels)) if cj != n_channels else parts[j, 0]
            mean = term.clone() if mean is None else mean + term
            smax = parts[j, 1].clone() if smax is None else torch.maximum(smax, parts[j, 1])
        if x.shape[0] > 0:
            gate_fn(x, torch.stack((mean, smax)), opacity_of_frame[f])
        if opacity_plane is not None and p0 <= opacity_plane < p0 + n:
            view[opacity_plane - p0].copy_(opacity_of_frame[f])


class PipelinedExchange:
    """The camera-frame exchange taken OFF a step's critical path: step k's two collectives run on a communication
    stream while step k + 1 pools and renders; the caller receives step k's complete fused grid one call later.

        pipe = PipelinedExchange(plan, rank, device, (Y, X))
        for every step k:
            pool the owned camera-frames of step k into pipe.pool_target(f)
            full_prev = pipe.submit()        # starts step k's exchange; -> step k - 1's fused grid (None at k = 0)
            ... consume full_prev (HOA-3, BEV encoder) before the NEXT submit ...
        full_last = pipe.flush()

    Why: inside a step the exchange is latency (reduce_scatter + all_gather over xGMI: ~ 0.1-0.2 ms by the cost model
    of DESIGN 6) that nothing of the SAME step can hide — the pooled BEV is the last thing the poolings produce and the
    first thing HOA-3 needs — so a strong-scaled step is pools / G + exchange + HOA-3 and barely beats one GPU.  Across
    steps the dependency is gone: a rank's throughput becomes max(own compute, exchange) at the price of one step of
    latency on the fused grid (renders and HOA-1/2 of step k are returned with it).  Two buffer sets alternate; both
    ride on the same process groups.  On the CPU (gloo tests) the collectives complete inside ``submit``: same
    results, no overlap."""

    def __init__(self, plan, rank, device, plane_shape, dtype=torch.float32, collectives=None, first=None):
        """``first``: an existing ``BevExchange`` of this plan to use as buffer set 0 (its process groups are shared)."""
        if first is None:
            first = BevExchange(plan, rank, device, plane_shape, dtype=dtype, collectives=collectives)
        self.slots = [first, BevExchange(plan, rank, device, plane_shape, dtype=dtype, share=first)]
        if first.touched:
            self.slots[1].adopt_touched(first)
        self.plan, self.rank, self.device = plan, rank, torch.device(device)
        self.k = 0
        self._pending = None                     # (full, event or None) of the step whose exchange is in flight
        self._comm = torch.cuda.Stream(self.device) if self.device.type == 'cuda' else None

    @property
    def active(self):
        return self.slots[0].active

    @property
    def current(self):
        return self.slots[self.k % 2]

    def set_touched(self, touched):
        """World collective (once): the wedge-sparse step 1 for both buffer sets."""
        self.slots[0].set_touched(touched)
        self.slots[1].adopt_touched(self.slots[0])

    def pool_target(self, f):
        return self.current.pool_target(f)

    def submit(self, between=None):
        """Start the exchange of the step just pooled (its buffers must not be written again before the call after
        next); -> the previous step's complete fused grid ``(n_frames, P, Y, X)`` ordered on the caller's stream, or
        None for the first step.  The returned tensor is overwritten by the exchange two submits later.
        ``between(ex)``: work on this rank's finished plane blocks between the two collectives (``gate_blocks``), issued on
        the communication stream."""
        ex = self.current

        def exchange():
            ex.finish_reduce(ex.start())
            if between is not None:
                between(ex)
            return ex.gather()
        if self._comm is not None:
            cur = torch.cuda.current_stream(self.device)
            self._comm.wait_stream(cur)                       # the poolings (and HOA-1/2) of this step
            with torch.cuda.stream(self._comm):
                full = exchange()
                done = torch.cuda.Event()
                done.record(self._comm)
        else:
            full, done = exchange(), None
        prev, self._pending = self._pending, (full, done)
        self.k += 1
        return self._take(prev)

    def flush(self):
        """-> the fused grid of the last submitted step (None if there is none)."""
        prev, self._pending = self._pending, None
        return self._take(prev)

    def _take(self, pending):
        if pending is None:
            return None
        full, done = pending
        if done is not None:
            torch.cuda.current_stream(self.device).wait_event(done)
        return full
