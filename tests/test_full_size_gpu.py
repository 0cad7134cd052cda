"""GPU: the full-size configurations of BASELINE.json that the small parity tests do not reach —
configs[2] (6 cams 256x704, 200x200 BEV: 520 000 Gaussians) under BOTH render-camera conventions (the
reference's own set-up, view_transformer_ocrf.py:1135-1152, and the corrected one the bench times) and
configs[4] (512x1408 input: 32x88 feature map, 88x32 tile grid, 8 frames) — against the C oracle where it
finishes in seconds, through size-independent properties elsewhere."""
import numpy as np
import pytest
import torch

from ocrfdet_amd import hotpath, synthetic
from tests import helpers
from tests.test_rasterize_gpu import AMBIGUITY, _compare

pytestmark = pytest.mark.gpu
CFG2 = 'cfg2_6cam_2frame_bev200x200_render_hoa'
CFG4 = 'cfg4_6cam_8frame_512x1408_bev200x200'


def _one_frame(key):
    return synthetic.PathConfig(**{**synthetic.CONFIGS[key].__dict__, 'n_frames': 1, 'hoa': False})


def _render_vs_oracle(hp, oracle_lib, views, max_outlier_frac=2e-5):
    g, rc = hp.gauss, hp.render_cams
    H, W = hp.cfg.input_size
    xyz = hp.voxel_xyz[0].reshape(-1, 3)
    got = hp.render(want_n_contrib=True)[0]
    torch.cuda.synchronize()
    assert int(got['status'].item()) & 1 == 0
    rendered = []
    for v in views:
        want = oracle_lib.rasterize_forward(xyz.cpu().numpy(), g['rgb'].cpu().numpy(), g['opacity'].cpu().numpy(),
                                            g['scales'].cpu().numpy(), g['rotations'].cpu().numpy(),
                                            rc['vm'][v].cpu().numpy(), rc['pm'][v].cpu().numpy(), rc['tfx'][v], rc['tfy'][v],
                                            H, W, np.zeros(3, np.float32), ambiguity=AMBIGUITY)
        one = {k: got[k][v:v + 1].cpu().numpy() for k in ('color', 'depth', 'final_T', 'n_contrib', 'radii')}
        one['tiles_touched'] = want['tiles_touched'][None]        # not returned by the batched call: radii pin the rects
        _compare(want, one, H, W, max_outlier_frac, label=f'{hp.cfg.name} view {v} ({hp.render_convention})')
        rendered.append(want['num_rendered'])
    return rendered


@pytest.mark.parametrize('convention', ['reference', 'corrected'])
def test_cfg2_render_both_camera_conventions(cuda, oracle_lib, convention):
    hp = hotpath.HotPath(_one_frame(CFG2), cuda)
    r = synthetic.rig(hp.cfg.n_cams, hp.cfg.input_size, hp.batch)
    hp._prepare_render(r, convention=convention)
    n = _render_vs_oracle(hp, oracle_lib, views=(1, 4))
    # the reference's unscaled intrinsics give a narrow field of view: few tile instances; corrected: millions
    assert (max(n) < 3_000_000) if convention == 'reference' else (min(n) > 1_000_000)


@pytest.mark.parametrize('gaussians', ['stress', 'objects'])
def test_cfg2_render_on_the_other_gaussian_sets(cuda, oracle_lib, gaussians):
    """SURVEY 8d's stress set and the object-centric set (synthetic.grid_gaussians) at full size: the planned render of
    BOTH parameter sets a step alternates between equals the per-call pipeline bit for bit (the head of the second call lags:
    it was set by the other set), and two views of the per-call render lie inside the C oracle's threshold-ambiguity map."""
    hp = hotpath.HotPath(_one_frame(CFG2), cuda, gaussians=gaussians, alternate=True)
    for phase in (0, 1, 0):
        hp.set_phase(phase)
        planned = hp.render()[0]
        per_call = hotpath.HotPath.render(hp, want_n_contrib=True)[0]
        torch.cuda.synchronize()
        for k in ('color', 'depth', 'final_T'):
            assert torch.equal(planned[k], per_call[k]), (gaussians, phase, k)
    hp.check_render_plans()
    if gaussians == 'objects':
        assert float(planned['final_T'].max()) > 0.05      # pixels that never saturate: nothing ends a tile's list early
    # every pixel off by more than 1e-4 must lie in the oracle's threshold-ambiguity map (_compare).  Their NUMBER is capped
    # in proportion to the decisions a pixel takes: ~ 80 blended records per pixel on the init set (cap 2e-5: 3 px), ~ 750 on
    # the object-centric one, whose map holds ~ 12 000 of 180 224 pixels (9 and 14 px flipped on the two views: cap 2e-4)
    _render_vs_oracle(hp, oracle_lib, views=(1, 4), max_outlier_frac=2e-4 if gaussians == 'objects' else 5e-5)


def test_replayed_step_follows_a_scene_that_turns_object_centric_and_back(cuda):
    """The recorded step (one host call) re-reads the render plans' host-visible hint at EVERY replay: a scene whose opacity
    field turns object-centric IN PLACE (same tensors, new values — what a network's heads do over a training run) is handed to
    the second pass once, then rendered by the build of the blend that reads per-call compacted candidates, and goes back to the
    plain build when the field saturates again; every step's renders equal the per-call pipeline's bit for bit."""
    cfg = _one_frame(CFG2)
    hp = hotpath.HotPath(cfg, cuda)                                   # init set, planned, one host call
    ref = hotpath.HotPath(cfg, cuda, render_mode='per_call', one_call=False)
    depth, feat = hp.make_inputs(seed=1)
    xyz = hp.voxel_xyz[0].reshape(-1, 3).cpu().numpy()
    sets = {k: synthetic.grid_gaussians(k, xyz, seed=7) for k in ('init', 'objects')}
    hints = []
    for kind in ('init', 'init', 'init', 'objects', 'objects', 'objects', 'objects', 'init', 'init', 'init'):
        for h in (hp, ref):
            for k, v in sets[kind].items():
                h.frame_gauss[0][k].copy_(torch.from_numpy(v).to(cuda))
        for plan, f0, nf, gg in hp._plans():
            for k in ('rgb', 'opacity', 'scales', 'rotations'):
                gg[k].copy_(torch.stack([hp.frame_gauss[b][k] for b in range(f0, f0 + nf)]))
        got = hp.step(depth, feat)[2][0]
        want = ref.render()[0]
        torch.cuda.synchronize()
        for k in ('color', 'depth', 'final_T'):
            assert torch.equal(got[k], want[k]), (kind, k, len(hints))
        hints.append(int(hp.render_plans[0][0]._hint.item()))
    assert hp._compiled, getattr(hp, 'one_call_refused', None)
    assert hints[:3] == [0, 0, 0] and hints[3:7] == [1, 1, 1, 1] and hints[-2:] == [0, 0], hints
    hp.check_render_plans()


def test_a_recording_that_misses_a_launch_is_refused(cuda):
    """The recorder logs the library's entry points only.  A step with a torch kernel of its own in it (here: one added to the
    pooled LSS volume after the fact) would replay WITHOUT that kernel: the first replay is held to the step issued call by
    call, the recording is dropped and the step stays call by call — still with the right values."""
    class Leaky(hotpath.HotPath):
        def _step_eager(self, depth, feat, rec=None):
            out = super()._step_eager(depth, feat, rec)
            out[0].add_(1.0)                                          # a launch the recorder cannot see
            return out

    cfg = _one_frame(CFG2)
    hp, ref = Leaky(cfg, cuda), hotpath.HotPath(cfg, cuda, one_call=False)
    depth, feat = hp.make_inputs(seed=1)
    want = ref.step(depth, feat)[0] + 1.0
    for _ in range(4):
        got = hp.step(depth, feat)[0]
        torch.cuda.synchronize()
        assert torch.equal(got, want)
    assert not hp._compiled and 'did not reproduce' in hp.one_call_refused


def test_cfg4_rank_triples_pools_and_render_at_512x1408(cuda, oracle_lib):
    cfg = _one_frame(CFG4)
    assert cfg.feat_hw == (32, 88)
    hp = hotpath.HotPath(cfg, cuda)                                                       # the default back ends: panel kernels
    assert hp.lss_pool_backend == 'panel' and hp.ht_pool_backend == 'panel'
    depth, feat = hp.make_inputs(seed=2)
    lss, ht, rendered = hp.step(depth, feat)[:3]
    torch.cuda.synchronize()
    X, Y, Z = cfg.bev_xyz
    # pools against the C oracle (LSS: 2 M points)
    d, f = depth.cpu().numpy(), feat.cpu().numpy()
    for name, plan, got in (('lss', hp.lss, lss), ('ht', hp.ht, ht)):
        want = oracle_lib.bev_pool_v2(d, f, plan.ranks_depth.cpu().numpy(), plan.ranks_feat.cpu().numpy(),
                                      plan.ranks_bev.cpu().numpy(), plan.bev_shape, plan.starts.cpu().numpy(),
                                      plan.lengths.cpu().numpy())
        want = np.concatenate([want[:, :, z] for z in range(want.shape[2])], 1)
        np.testing.assert_allclose(got.cpu().numpy(), want, rtol=1e-4, atol=1e-4, err_msg=name)
    # the per-step HIP index preparation at this shape gives the same pooled BEVs bit for bit
    # (that path pools with the tile kernel: bit for bit against the cached ranks on the tile kernel, 1e-5 against the panels)
    no_render = synthetic.PathConfig(**{**cfg.__dict__, 'render': False})
    hp2 = hotpath.HotPath(no_render, cuda, index_prep_mode='per_step')
    lss2, ht2 = hp2.step(depth, feat)[:2]
    hp3 = hotpath.HotPath(no_render, cuda, lss_pool_backend='tile', ht_pool_backend='tile')
    lss3, ht3 = hp3.step(depth, feat)[:2]
    assert torch.equal(lss2, lss3) and torch.equal(ht2, ht3)
    torch.testing.assert_close(lss2, lss, rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(ht2, ht, rtol=1e-5, atol=1e-5)
    # render: 88 x 32 tiles, one view against the oracle
    H, W = cfg.input_size
    assert rendered[0]['color'].shape == (6, 3, H, W)
    _render_vs_oracle(hp, oracle_lib, views=(0,))


def test_cfg4_eight_frames_step_properties(cuda):
    """The whole configs[4] step (8 frames x 6 cams, 48 rendered views): finite, and a frame agrees bit for bit with a
    one-frame run of that frame (frames ride along as batch entries; every frame has its own ego pose and its own
    Gaussian parameters, synthetic.ego_motion)."""
    cfg = synthetic.PathConfig(**{**synthetic.CONFIGS[CFG4].__dict__, 'hoa': False})
    hp = hotpath.HotPath(cfg, cuda)
    depth, feat = hp.make_inputs(seed=4)
    lss, ht, rendered = hp.step(depth, feat)[:3]
    torch.cuda.synchronize()
    assert lss.shape[0] == 8 and len(rendered) == 8
    assert torch.isfinite(lss).all() and torch.isfinite(ht).all()
    one = hotpath.HotPath(_one_frame(CFG4), cuda, frame_offset=3)
    l1, h1, r1 = one.step(depth[3:4].contiguous(), feat[3:4].contiguous())[:3]
    assert torch.equal(l1[0], lss[3]) and torch.equal(h1[0], ht[3])
    assert torch.equal(r1[0]['color'], rendered[3]['color']) and torch.equal(r1[0]['depth'], rendered[3]['depth'])
    for k, o in enumerate(rendered):
        assert torch.isfinite(o['color']).all() and float(o['final_T'].min()) >= 0.0
        assert k == 0 or not torch.equal(o['color'], rendered[0]['color'])     # another pose, other Gaussians
    hp.check_render_plans()
    # the planned renders of the step equal the per-call pipeline's
    per_call = hotpath.HotPath(cfg, cuda, render_mode='per_call').render()
    for a, b in zip(rendered, per_call):
        assert torch.equal(a['color'], b['color']) and torch.equal(a['depth'], b['depth']) and \
            torch.equal(a['final_T'], b['final_T'])


def _pool_oracle(oracle_lib, plan, d, f):
    want = oracle_lib.bev_pool_v2(d, f, plan.ranks_depth.cpu().numpy(), plan.ranks_feat.cpu().numpy(),
                                  plan.ranks_bev.cpu().numpy(), plan.bev_shape, plan.starts.cpu().numpy(),
                                  plan.lengths.cpu().numpy())
    return np.concatenate([want[:, :, z] for z in range(want.shape[2])], 1)


def test_cfg2_step_as_benched_against_per_call_renders_and_oracles(cuda, oracle_lib):
    """The step ``bench.py`` times at N = 1, built exactly as it builds it (``HotPath(cfg2)`` with every default: both
    frames rendered through ONE fused 12-view plan with ``item_view`` — head kernel, first pass of the persistent blend on
    2.75 workgroups per CU of the side stream over the plan's candidate lists, extent check, second pass —, both poolings
    on the panel kernel behind one cell-weight pre-pass, HOA-1/2 as the two + six latency kernels, HOA-3 as two; the
    second step replayed by ONE host call) — every output of it: renders bit-equal to the per-call pipeline, pooled BEVs
    within 1e-4 of the C oracle, opacity BEV and gated BEV against the numpy HOA oracle, plan extent check clean."""
    from oracle import hoa as ohoa
    cfg = synthetic.CONFIGS[CFG2]
    hp = hotpath.HotPath(cfg, cuda)
    depth, feat = hp.make_inputs(seed=0)
    for _ in range(2):                                   # the second step reuses every persistent buffer
        lss, ht, rendered, gated, ob = hp.step(depth, feat)
    torch.cuda.synchronize()
    hp.check_render_plans()
    assert len(hp.render_plans) == 1 and hp.render_plans[0][2] == 2       # one plan, two frames: the fused form
    # renders: the planned path of the step == the per-call pipeline, bit for bit, both frames, all six views
    per_call = hotpath.HotPath(cfg, cuda, render_mode='per_call', overlap=False).render()
    assert len(rendered) == 2 and rendered[0]['color'].shape == (6, 3, *cfg.input_size)
    for f in range(2):
        for k in ('color', 'depth', 'final_T'):
            assert torch.equal(rendered[f][k], per_call[f][k]), f'frame {f}: planned {k} differs from the per-call render'
    assert not torch.equal(rendered[0]['color'], rendered[1]['color'])
    # pooled BEVs (both on the panel kernel, the default) vs the C oracle
    d, ft = depth.cpu().numpy(), feat.cpu().numpy()
    for name, plan, got in (('lss', hp.lss, lss), ('ht', hp.ht, ht)):
        err = float(np.abs(got.cpu().numpy() - _pool_oracle(oracle_lib, plan, d, ft)).max())
        assert err <= 1e-4, f'{name}: {err}'
    # HOA-1/2 -> opacity BEV, HOA-3 -> gated BEV vs the numpy oracle (per sample, as the reference loops)
    X, Y, _ = cfg.bev_xyz
    Zh = cfg.num_height
    p = {f'{pre}.{k}': v.detach().cpu().numpy() for pre, m in hp.hoa_mods.items() for k, v in m.state_dict().items()}
    errs = []
    for b in range(hp.batch):
        op = hp.frame_gauss[b]['opacity'].cpu().numpy().astype(np.float32)
        oa, _ = ohoa.hoa1(op, hp.alpha_lidar[b:b + 1].cpu().numpy(), p, Zh, Y, X, prefix='dca')
        want_ob = ohoa.opacity_voxel_to_bev(oa, hp.bev_pos1[b:b + 1].cpu().numpy(), p, 'v2b')
        want_mask = ohoa.opacity_mask(ht[b:b + 1].cpu().numpy(), want_ob, p, 'mask')
        e_ob = float(np.abs(ob[b:b + 1].cpu().numpy() - want_ob).max())
        e_g = float(np.abs(gated[b:b + 1].cpu().numpy() - ht[b:b + 1].cpu().numpy() * want_mask).max())
        errs.append((e_ob, e_g))
        assert e_ob <= 1e-5 and e_g <= 1e-5, (b, e_ob, e_g)
    print('cfg2 benched step: max|opacity_bev err|, max|gated err| per frame', errs)


def test_cfg2_step_options_give_the_same_outputs(cuda):
    """Pooling back ends that were measured against the default: the panel poolings within the
    summation-order tolerance of the LSS grid (the HT grid: bit-identical to the MFMA form).  And the step issued by ONE
    host call (``ocrf_hotpath_step``: its library calls recorded once, replayed from C) equals the step issued call by
    call, bit for bit — after EVERY input of the step has been changed in place, so that a replay that missed a launch
    (or a torch op between the library's own) would show stale values."""
    cfg = synthetic.CONFIGS[CFG2]
    ref_hp = hotpath.HotPath(cfg, cuda, one_call=False)
    depth, feat = ref_hp.make_inputs(seed=1)
    ref = ref_hp.step(depth, feat)
    torch.cuda.synchronize()
    hp = hotpath.HotPath(cfg, cuda, lss_pool_backend='panel', ht_pool_backend='panel', one_call=False)
    for _ in range(2):
        out = hp.step(depth, feat)
    torch.cuda.synchronize()
    assert hp.lss.mfma_plan is not None and hp.ht.mfma_plan is not None
    torch.testing.assert_close(out[0], ref[0], rtol=1e-5, atol=1e-5)        # LSS: tile kernel vs panel plan (another order)
    assert torch.equal(out[1], ref[1])                                        # HT: the MFMA form's bits
    assert torch.equal(out[3], ref[3]) and torch.equal(out[4], ref[4])
    del hp

    one = hotpath.HotPath(cfg, cuda)                   # one_call=True is the default
    d1, f1 = depth.clone(), feat.clone()
    one.step(d1, f1)                                   # issued call by call (builds plans and scratch)
    one.step(d1, f1)                                   # recorded (issued call by call once more, its library calls logged)
    one.step(d1, f1)                                   # replayed: one host call
    assert one._compiled and next(iter(one._compiled.values()))[0].n_calls >= 7
    # new values everywhere, same tensors: inputs of the poolings, of the renders, of HOA
    d2, f2 = ref_hp.make_inputs(seed=5)
    d1.copy_(d2), f1.copy_(f2)
    X, Y, _ = cfg.bev_xyz
    for hpx in (one, ref_hp):
        g = torch.Generator(device='cpu').manual_seed(11)
        for fg in hpx.frame_gauss:
            fg['opacity'].copy_((torch.rand(fg['opacity'].shape, generator=g) * 0.3 + 0.3).to(cuda))
            fg['rgb'].copy_(torch.rand(fg['rgb'].shape, generator=g).to(cuda))
            fg['scales'].copy_((torch.rand(fg['scales'].shape, generator=g) * 0.1 + 0.7).to(cuda))
        hpx.alpha_lidar.copy_(torch.rand(hpx.alpha_lidar.shape, generator=g).to(cuda))
        # the step reads the STACKED parameter tensors of its plans and the flat opacity volume of HOA-1: in place too
        for plan, f0, nf, gg in hpx._plans():
            for k in ('rgb', 'opacity', 'scales', 'rotations'):
                gg[k].copy_(torch.stack([hpx.frame_gauss[b][k] for b in range(f0, f0 + nf)]))
        hpx._opac_flat.copy_(torch.stack([fg['opacity'].view(cfg.num_height, Y, X) for fg in hpx.frame_gauss]).reshape(-1, 1))
    want = ref_hp.step(d2, f2)
    got = one.step(d1, f1)
    torch.cuda.synchronize()
    for i in (0, 1, 3, 4):
        assert torch.equal(got[i], want[i]), i
    for fr_g, fr_w in zip(got[2], want[2]):
        for k in ('color', 'depth', 'final_T'):
            assert torch.equal(fr_g[k], fr_w[k]), k
    one.check_render_plans()


@pytest.mark.parametrize('kw', [dict(render_mode='per_call'), dict(plan_rebuild='per_step')])
def test_per_sample_step_by_one_host_call_equals_the_step_call_by_call(cuda, kw):
    """The per-SAMPLE step (index preparation from the calibration on the device, pooling on device-side counts, the
    per-call render or the plan rebuilt per step; four HIP streams) replayed by ONE host call equals the same step issued
    call by call, bit for bit — after depth / features, the calibration (translations) and the Gaussian parameters were
    changed in place, so that a replay that missed a launch would show stale values."""
    cfg = synthetic.CONFIGS[CFG2]
    mk = lambda one: hotpath.HotPath(cfg, cuda, index_prep_mode='per_step', device_geometry=True, one_call=one, **kw)  # noqa: E731
    ref, one = mk(False), mk(True)
    depth, feat = ref.make_inputs(seed=2)
    d1, f1 = depth.clone(), feat.clone()
    for _ in range(3):
        one.step(d1, f1)                               # call by call, recorded, replayed
    assert one._compiled, getattr(one, 'one_call_refused', None)
    before = [t.clone() for t in ref.step(depth, feat)[:2]]
    d2, f2 = ref.make_inputs(seed=6)
    d1.copy_(d2), f1.copy_(f2)
    for hpx in (one, ref):
        hpx._calib_dev[1].add_(0.05)                   # every camera 5 cm further along x, y, z: other rank vectors
        g = torch.Generator(device='cpu').manual_seed(3)
        for fg in hpx.frame_gauss:
            fg['opacity'].copy_((torch.rand(fg['opacity'].shape, generator=g) * 0.3 + 0.3).to(cuda))
            fg['rgb'].copy_(torch.rand(fg['rgb'].shape, generator=g).to(cuda))
        if hpx.render_plans:
            for plan, f0, nf, gg in hpx.render_plans:
                for k in ('rgb', 'opacity'):
                    gg[k].copy_(torch.stack([hpx.frame_gauss[b][k] for b in range(f0, f0 + nf)]))
        if getattr(hpx, '_sets', None):                  # the per-call render of the batch reads the stacked sets
            for k in ('rgb', 'opacity'):
                hpx._sets[k].copy_(torch.stack([fg[k] for fg in hpx.frame_gauss]))
        X, Y, _ = cfg.bev_xyz
        hpx._opac_flat.copy_(torch.stack([fg['opacity'].view(cfg.num_height, Y, X) for fg in hpx.frame_gauss]).reshape(-1, 1))
    want = ref.step(d2, f2)
    got = one.step(d1, f1)
    torch.cuda.synchronize()
    assert not torch.equal(want[0], before[0]) and not torch.equal(want[1], before[1])      # the inputs did change the step
    for i in (0, 1, 3, 4):
        assert torch.equal(got[i], want[i]), (i, float((got[i] - want[i]).abs().max()), int((got[i] != want[i]).sum()),
                                              bool(torch.equal(got[i], before[i])) if i < 2 else None)
    for fr_g, fr_w in zip(got[2], want[2]):
        for k in ('color', 'depth', 'final_T'):
            assert torch.equal(fr_g[k], fr_w[k]), k
    one.check_render_plans(), ref.check_render_plans()
