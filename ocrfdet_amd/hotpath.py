"""The hot path as one object: cached index preparation + LSS pool + HT pool (+ render + HOA).

This is the build's analogue of ``OcRFViewTransformerFull.view_transform_core``
(mmdet3d/models/necks/view_transformer_ocrf.py:1040-1201) restricted to the custom-kernel stages,
with the rank vectors pre-computed once per calibration as the reference's ``accelerate=True``
path intends (``pre_compute``, view_transformer_ocrf.py:854-866).  ``bench.py``,
``__graft_entry__.smoke()`` and the sharded runner drive it; frames are treated as extra batch
entries because they are independent until the channel concat (detectors/ocrfdet.py:274).
"""
import contextlib
import math

import numpy as np
import torch

from . import _lib, bevpool, gaussian_renderer, hoa, index_prep, raster_plan, synthetic
from .diff_gaussian_rasterization import pack_cameras, rasterize_sets, rasterize_views


class PoolPlan:
    """Rank vectors of one bev_pool_v2 call, resident on the device."""

    def __init__(self, ranks_bev, ranks_depth, ranks_feat, starts, lengths, bev_shape):
        self.ranks_bev, self.ranks_depth, self.ranks_feat = ranks_bev, ranks_depth, ranks_feat
        self.starts, self.lengths = starts, lengths
        self.bev_shape = tuple(int(v) for v in bev_shape)      # (B, Z, Y, X, C)
        self.device_plan = None                                # bevpool.DevicePoolPlan, built on first use
        self.mfma_plan = None                                  # bevpool.MfmaPoolPlan, built on first use

    @property
    def n_points(self):
        return int(self.ranks_bev.numel())

    @property
    def n_intervals(self):
        return int(self.starts.numel())

    def algorithmic_bytes(self, depth_numel, feat_numel):
        """4*(depth + feat + 3*Np + 2*Nv + B*Z*Y*X*C)."""
        out = 1
        for v in self.bev_shape:
            out *= v
        return 4 * (depth_numel + feat_numel + 3 * self.n_points + 2 * self.n_intervals + out)


_STREAMS = {}


def shared_stream(device, role):
    """One side HIP stream per (device, role) for the whole process: every HotPath reuses the same two streams
    ('render', 'prep') instead of creating its own — the streams of a process share a few hardware queues in creation
    order, and which queue a stream lands on decides whether it really runs beside the main stream."""
    key = (torch.device(device).index or 0, role)
    if key not in _STREAMS:
        _STREAMS[key] = torch.cuda.Stream(device)
    return _STREAMS[key]


class HotPath:
    def __init__(self, cfg, device, cams=None, index_prep_mode='cached', overlap=True, device_geometry=False,
                 render_mode='planned', render_guard='host', frame_motion=True, frame_offset=0, plan_margin=1.25,
                 ht_pool_backend=None, fuse_frames='auto', render_streams=1, blend_workgroups='auto',
                 lss_pool_backend=None, lss_mfma_group=2, plan_rebuild='never', hoa_first=None, one_call=True,
                 gaussians='init', alternate=None):
        """``cams``: optional list of camera indices this instance owns (camera sharding).
        ``index_prep_mode``: 'cached' — rank vectors computed once per calibration, the reference's
        ``accelerate=True`` intent; 'per_step' — recomputed inside every ``step()`` by the HIP index
        preparation (csrc/index_prep.hip), what the reference does with ``accelerate=False``.
        ``render_mode``: 'planned' — the render's calibration-only front end (cull, depth order, projected centres)
        is cached per frame like the rank vectors (``raster_plan.RasterPlan``; the Gaussian means are the fixed voxel
        grid, view_transformer_ocrf.py:651-673); 'per_call' — recomputed by every render (``rasterize_views``).
        ``render_guard``: 'host' (the plan's extent bound is checked by ``check_render_plans()``) or 'device'.
        ``plan_rebuild``: 'never' — a render plan lives as long as its calibration; 'per_step' — every step rebuilds it
        on the device from the sample's camera block before rendering (``RasterPlan.rebuild``: no host read), what a
        per-SAMPLE pose costs: the reference builds the render cameras from the dataloader's c2w for every sample
        (view_transformer_ocrf.py:1140-1152, detectors/ocrfdet.py:215-223).
        ``frame_motion``: batch entry b is frame ``frame_offset + b`` of a multi-frame sample with its own ego pose
        (``synthetic.ego_motion``) and its own Gaussian parameters, instead of every frame repeating frame 0.
        ``lss_pool_backend`` / ``ht_pool_backend``: 'tile' | 'mfma' | 'panel'; None = 'panel', the latency kernel of
        csrc/bev_pool_panel.hip: 13-30 % faster than tile / mfma alone on the device, and — since round 5, with the host
        out of the step's way and the blend on 2.75 workgroups per CU — also beside the blend (cfg2 step 0.200 vs 0.208 /
        0.212 ms with tile LSS / mfma HT, tools/sweep_r5_step.py).
        ``hoa_first``: HOA-1/2 before the poolings (None: with the per-call render or the per-step index preparation).
        ``gaussians``: the synthetic Gaussian parameter set (``synthetic.grid_gaussians``): 'init' | 'stress' | 'objects'.
        ``alternate``: draw TWO parameter sets per frame and let the caller switch between them from step to step
        (``set_phase``; default: only for 'objects') — new parameters every step, as a network's heads produce them: the
        planned render's adaptive head then always works from the OTHER set's reach.
        ``one_call``: after its first, recorded, issue a step is ONE host call (``ocrf_hotpath_step``: the library calls
        of the step replayed from C, ``_lib.StepRecorder``) instead of ~14 ctypes calls + torch stream / event calls
        (225 -> ~60 us of host work at cfg2).  The step's outputs are then the SAME tensors every step (overwritten by
        the next one, like a replayed graph's), and ``depth`` / ``feat`` have to be the tensors of the recorded call
        (another pair is recorded anew)."""
        self.cfg, self.device = cfg, torch.device(device)
        assert render_mode in ('planned', 'per_call') and render_guard in ('host', 'device')
        assert plan_rebuild in ('never', 'per_step')
        self.plan_rebuild = plan_rebuild
        assert gaussians in synthetic.GAUSSIAN_SETS
        self.gaussians = gaussians
        self.plan_bins = (4, 2)                # candidate lists of the render plans: bins of 4 x 2 tile pairs (64 x 64 px)
        self.alternate = (gaussians == 'objects') if alternate is None else bool(alternate)
        self.phase = 0
        self.render_mode, self.render_guard, self.plan_margin = render_mode, render_guard, float(plan_margin)
        self.frame_motion, self.frame_offset = bool(frame_motion), int(frame_offset)
        # 'mfma': the HT pooling (cached ranks) as per-tile MFMA panels (csrc/bev_pool_mfma.hip: 26 vs 32 us at cfg2);
        # 'tile': the VALU tile kernel for both poolings.  The LSS ranks keep the tile kernel (its heavy tiles — a
        # 3.2 m block beside the rig collects thousands of rows — make the per-tile MFMA chain the launch's tail).
        self.overlap = bool(overlap) and self.device.type == 'cuda'
        if lss_pool_backend is None:
            lss_pool_backend = 'panel'
        if ht_pool_backend is None:
            ht_pool_backend = 'panel'
        assert ht_pool_backend in ('mfma', 'tile', 'panel') and lss_pool_backend in ('mfma', 'tile', 'panel')
        self.lss_panel_unit_cost = 8.0
        self.ht_pool_backend = ht_pool_backend
        self.lss_pool_backend, self.lss_mfma_group = lss_pool_backend, int(lss_mfma_group)
        self.hoa_first = hoa_first
        self.one_call = bool(one_call) and self.device.type == 'cuda'
        self._compiled = {}                    # key -> (_lib.CompiledStep, outputs, memory pool, scratch hold) of a recorded step
        self._one_call_ok = True
        self._warm_keys = set()
        # planned renders: consecutive frames in one plan / one launch pair.  'auto': when ALL frames fit one plan
        # (<= 32 views; cfg2: 0.300 -> 0.285 ms — one update -> blend hand-over and one drain of the persistent grid
        # less); with more frames a launch pair per frame is faster (cfg4: 3.08 vs 3.18 ms in groups of five)
        self.fuse_frames = fuse_frames
        self.render_streams = max(1, int(render_streams))      # side HIP streams the frames' renders are dealt over
        # size of the planned blend's persistent grid per frame (int, or a sequence with one entry per frame; 0 = what
        # the device holds).  'auto': beside the main chain (overlap) the VALU-bound blend takes 3.5 workgroups per CU
        # and leaves the other wave slots to the latency-bound poolings / HOA of the main stream (cfg2: 0.344 -> 0.30 ms)
        self.blend_workgroups = blend_workgroups
        self._busy = torch.zeros(1, dtype=torch.int32, device=self.device) if self.device.type == 'cuda' else None
        self.cams = list(range(cfg.n_cams)) if cams is None else list(cams)
        self.batch = cfg.batch * cfg.n_frames                  # frames ride along as batch entries
        assert index_prep_mode in ('cached', 'per_step')
        self.index_prep_mode = index_prep_mode
        # per-step mode only: the per-camera calibration algebra on the GPU too (ocrf_geometry_blocks; ~1 ulp from
        # the host formulation, so a 1e-5 fraction of border points may change cell) instead of on the host
        self.device_geometry = bool(device_geometry)
        # the renders are independent of the poolings: with ``overlap`` they run on ONE side HIP stream
        # beside the pools + HOA of the main stream (the blend is VALU-bound with a ragged tail, the pools
        # and the small HOA kernels are latency / L2-bound: they interleave).  One stream per frame was
        # worse: two blends at once slow each other more than the overlap returns (0.66 vs 0.60 ms)
        self._side = []
        self._prep_stream = self._prep_stream2 = None
        self._prepare()

    def _prepare(self):
        cfg, dev = self.cfg, self.device
        r = synthetic.rig(cfg.n_cams, cfg.input_size, self.batch, frame_motion=self.frame_motion,
                          frame_offset=self.frame_offset)
        sel = self.cams
        g = {k: torch.from_numpy(r[k][:, sel] if r[k].ndim >= 3 and k != 'bda' else r[k]).to(dev)
             for k in ('rots', 'trans', 'intrins', 'post_rots', 'post_trans', 'bda', 'c2w')}
        self.geom = g
        X, Y, Z = cfg.bev_xyz
        Hf, Wf = cfg.feat_hw
        C = cfg.channels
        # Rank vectors of both branches from the HIP index preparation (csrc/index_prep.hip), which is pinned
        # bit-exactly to the reference's vectors at every configuration (tests/test_index_prep_gpu.py).  The same
        # formulas as torch ops ON THE GPU are not: torch's elementwise GPU kernels contract multiply-adds, and at
        # 512x1408 three pillar samples land on the other side of a .round() (view_transformer_ocrf.py:810).
        # The tiny per-camera 3x3 algebra runs on the host, as the same torch calls the reference makes.
        self._calib_host = [torch.from_numpy(r[k][:, sel] if k != 'bda' else r[k]) for k in
                            ('rots', 'trans', 'intrins', 'post_rots', 'post_trans', 'bda')]
        frustum = index_prep.create_frustum(cfg.grid['depth'], cfg.input_size, cfg.downsample)
        self._frustum_dev = frustum.to(dev).contiguous()
        self._ref_template = index_prep.get_reference_points_3d(Y, X, bs=1, num_points_in_pillar=cfg.num_height,
                                                                device='cpu')[0].to(dev).contiguous()
        self._grid = index_prep.grid_infos(cfg.grid)
        self._lss_bufs, self._ht_bufs = index_prep._RankBuffers(), index_prep._RankBuffers()
        self.lss, self.ht = self.prepare_indices_hip(sync=True)       # (host algebra: `_calib_dev` does not exist yet)
        # contiguous fp32 on the device: ocrf_geometry_blocks then reads THESE tensors (a conversion inside the per-step
        # path would be a torch kernel per step — one that a recorded step would not replay)
        self._calib_dev = [t.to(dev).float().contiguous() for t in self._calib_host]
        # cached plans must not alias the grow-only buffers the per-step preparation writes into
        for plan in (self.lss, self.ht):
            for name in ('ranks_bev', 'ranks_depth', 'ranks_feat', 'starts', 'lengths'):
                setattr(plan, name, getattr(plan, name).clone())
        # metric voxel centres (B, Zh, Y*X, 3): the Gaussian means of the render (ocrf_ht_project, bit-exact vs the
        # oracle of get_sampling_point's in-place scaling, view_transformer_ocrf.py:690-692)
        lidar2img, img_aug, _, _ = index_prep.get_projection(*self._calib_host)
        ht_block = index_prep.ht_camera_block(lidar2img, img_aug).to(dev)
        _, _, voxel = index_prep.ht_project_hip(self._ref_template, ht_block, self.batch, len(sel), list(cfg.pc_range),
                                                cfg.input_size, cfg.grid['depth'])
        self.voxel_xyz = voxel

        if cfg.render:
            self._prepare_render(r)
        if cfg.hoa:
            self._prepare_hoa()

    def _prepare_hoa(self, seed=0):
        """HOA blocks with seeded random-init weights of the reference architecture
        (view_transformer_ocrf.py:590-648) and the synthetic NeRF-branch alpha volume they consume."""
        cfg, dev = self.cfg, self.device
        X, Y, _ = cfg.bev_xyz
        torch.manual_seed(seed)
        self.hoa_mods = dict(
            dca=hoa.DeformableAttention2D(dim=cfg.num_height, dim_head=8, heads=1, dropout=0.1, downsample_factor=4,
                                          offset_scale=4, offset_groups=None, offset_kernel_size=6),
            v2b=hoa.OpacityVoxelToBEVConverter(input_channel=cfg.num_height), mask=hoa.ObatinOpacityMask())
        for m in self.hoa_mods.values():
            m.to(dev).eval()
        g = torch.Generator(device='cpu').manual_seed(seed)
        self.alpha_lidar = torch.rand(self.batch, cfg.num_height, Y, X, generator=g).to(dev)
        self._opac_flat = None
        self.bev_pos1 = (torch.randn(self.batch, 4, Y, X, generator=g) * 0.1).to(dev)

    @torch.no_grad()
    def hoa_opacity_bev(self):
        """HOA-1/2 (view_transformer_ocrf.py:1159-1161, 1196): opacity BEV (B,1,Y,X).  Independent of the pooled
        BEV (it reads the Gaussian opacities and the NeRF-branch alpha volume)."""
        cfg = self.cfg
        X, Y, _ = cfg.bev_xyz
        m = self.hoa_mods
        # every frame has its own opacity volume; the reference loops samples (:1090).  The (B*P, 1)
        # layout A_MLP hands over (:1130) is an INPUT of this stage: laid out once, not per step
        if self._opac_flat is None:
            self._opac_flats = [torch.stack([fg['opacity'].view(cfg.num_height, Y, X) for fg in fgs]).reshape(-1, 1).contiguous()
                                for fgs in self.frame_gauss_sets]
            self._opac_flat = self._opac_flats[self.phase]
        oa = hoa.hoa1(m['dca'], self._opac_flat, self.alpha_lidar, cfg.num_height, Y, X)
        return m['v2b'](oa, self.bev_pos1)

    @torch.no_grad()
    def hoa_step(self, geom_feat, opacity_bev=None):
        """HOA-1/2/3 (view_transformer_ocrf.py:1159-1161, 1196-1199): -> (gated BEV, opacity BEV)."""
        if opacity_bev is None:
            opacity_bev = self.hoa_opacity_bev()
        _, gated = self.hoa_mods['mask'].gate(geom_feat, opacity_bev)
        return gated, opacity_bev

    def _prepare_render(self, r, convention='corrected', seed=0):
        """Cameras + synthetic Gaussian parameters of the OcRF render, per frame.

        convention 'reference': the reference's own set-up, quirks included
        (view_transformer_ocrf.py:1135-1152: unscaled 1600x900 intrinsics with the network-input
        viewport, c2w fed as world->view); 'corrected': intrinsics scaled/cropped to the network
        input and a proper world->view transform (the headline of SURVEY.md 8d)."""
        cfg, dev = self.cfg, self.device
        H, W = cfg.input_size
        self.frame_cams = []
        n_sets = 2 if self.alternate else 1
        self.frame_gauss_sets = [[] for _ in range(n_sets)]      # [phase][frame] -> parameter dict
        P = self.voxel_xyz.shape[1] * self.voxel_xyz.shape[2]
        t = lambda a: torch.from_numpy(a).to(dev)  # noqa: E731
        for b in range(self.batch):
            vms, pms, tfx, tfy = [], [], [], []
            for n in self.cams:
                K = r['intrins'][b, n].astype(np.float64)
                c2w = r['c2w'][b, n].astype(np.float64)
                if convention == 'corrected':
                    s, crop = r['resize'], r['crop_h']
                    K = np.array([[K[0, 0] * s, 0, K[0, 2] * s], [0, K[1, 1] * s, K[1, 2] * s - crop], [0, 0, 1.0]])
                    w2c = np.linalg.inv(c2w)
                    c2w_arg = np.eye(4)
                    c2w_arg[:3, :3] = c2w[:3, :3]          # getWorld2View2 transposes R itself
                    c2w_arg[:3, 3] = w2c[:3, 3]
                else:
                    c2w_arg = c2w
                cam = gaussian_renderer.camera_from_calibration(K.astype(np.float32), c2w_arg.astype(np.float32), H, W)
                vms.append(cam['world_view_transform']), pms.append(cam['full_proj_transform'])
                tfx.append(math.tan(float(cam['FovX']) * 0.5)), tfy.append(math.tan(float(cam['FovY']) * 0.5))
            rc = dict(vm=torch.stack(vms).to(dev), pm=torch.stack(pms).to(dev), tfx=tfx, tfy=tfy)
            # the rig is fixed: pack the C ABI's camera block once instead of on every render call
            rc['packed'] = pack_cameras(rc['vm'], rc['pm'], tfx, tfy, H, W, dev)
            self.frame_cams.append(rc)
            # Gaussian parameters of frame (frame_offset + b): ranges of the reference's heads at seeded init
            # (SURVEY.md 8d probe); with frame_motion every frame draws its own
            xyz_np = self.voxel_xyz[b].reshape(-1, 3).cpu().numpy()
            for ph in range(n_sets):
                gs = synthetic.grid_gaussians(self.gaussians, xyz_np,
                                              seed + (self.frame_offset + b if self.frame_motion else 0) + 1000 * ph)
                self.frame_gauss_sets[ph].append({k: t(v) for k, v in gs.items()})
        self.frame_gauss = self.frame_gauss_sets[0]
        self.render_cams, self.gauss = self.frame_cams[0], self.frame_gauss[0]       # frame 0 (tests, tools)
        self.bg = torch.zeros(3, device=dev)
        self.render_convention = convention
        self.render_plans = None                 # raster_plan.RasterPlan per frame, built on first use
        self._opac_flat = None

    def _plans(self):
        """Render plans: every frame sees the SAME Gaussian means (the voxel grid) through its own cameras with its own
        parameter set, so consecutive frames share one plan of up to 32 views (``raster_plan.RasterPlan`` with one
        Gaussian set per frame): one update launch and one persistent blend launch for all of them instead of a pair
        per frame.  -> list of (plan, first frame, n frames, stacked parameters)."""
        if self.render_plans is None:
            H, W = self.cfg.input_size
            n_cam = len(self.cams)
            same = all(torch.equal(self.voxel_xyz[b], self.voxel_xyz[0]) for b in range(1, self.batch))
            # (round 5: also when the batch needs several plans — cfg4's 48 views as 30 + 18 instead of 8 x 6: every
            # launch less is a head kernel, an extent check and a blend tail less on the render stream, 2.156 -> 1.865 ms)
            fuse = True if self.fuse_frames == 'auto' else bool(self.fuse_frames)
            per = max(1, 32 // n_cam) if (same and fuse) else 1
            self.render_plans = []
            for f0 in range(0, self.batch, per):
                fr = list(range(f0, min(f0 + per, self.batch)))
                cams = torch.cat([self.frame_cams[b]['packed'] for b in fr])
                alts = [{k: torch.stack([fgs[b][k] for b in fr]).contiguous() for k in ('rgb', 'opacity', 'scales', 'rotations')}
                        for fgs in self.frame_gauss_sets]
                g = alts[0]
                # (the extent bound has to hold for every parameter set the plan will render)
                ext_sc = torch.cat([a['scales'].reshape(-1, 3) for a in alts])
                ext_rot = torch.cat([a['rotations'].reshape(-1, 4) for a in alts])
                # record capacity: what these cameras keep + 10 % — + 25 % when the plan is rebuilt per step for poses that
                # may keep more (a plan beyond its capacity is refused on the device: status bit 8, zero images; read
                # check_render_plans() before the images are consumed, or render with render_guard='device')
                plan = raster_plan.RasterPlan(self.voxel_xyz[f0].reshape(-1, 3), cams, H, W, scales=ext_sc,
                                              rotations=ext_rot, margin=self.plan_margin,
                                              headroom=1.25 if self.plan_rebuild == 'per_step' else 1.1,
                                              # a plan rebuilt per SAMPLE does not build candidate lists every step
                                              bins=None if self.plan_rebuild == 'per_step' else self.plan_bins)
                # item z = view z of the plan (frame-major), rendered with the parameter set of its frame
                for a in alts:
                    a['item_view'] = torch.arange(len(fr) * n_cam, dtype=torch.int32, device=self.device) if a is g else g['item_view']
                    a['cams'] = plan.cameras             # the sample's camera block: a per-step rebuild reads it in place
                g['alt'] = alts                          # [phase] -> parameter dict (alts[0] is g)
                self.render_plans.append((plan, f0, len(fr), g))
        return self.render_plans

    def check_render_plans(self):
        """Synchronising check of the render plans' extent bound (``render_guard='host'``): raises if a planned
        render since the last check was not valid."""
        for p in (getattr(self, 'render_plans', None) or []):
            p[0].check()

    def _render_per_call(self, b, want_n_contrib, tag):
        cfg, rc, g = self.cfg, self.frame_cams[b], self.frame_gauss[b]
        H, W = cfg.input_size
        xyz = self.voxel_xyz[b].reshape(-1, 3)
        return rasterize_views(xyz, g['rgb'], g['opacity'], g['scales'], g['rotations'], rc['vm'], rc['pm'], rc['tfx'],
                               rc['tfy'], H, W, self.bg, packed_cameras=rc['packed'], want_n_contrib=want_n_contrib,
                               workspace_tag=tag)

    def _use_busy(self):
        # the hint pays when a step has several blend launches (cfg4: two plans of 30 + 18 views; with one plan per frame
        # in round 3: 3.08 -> 2.70 ms); with ONE launch per step every extra workgroup would leave at once anyway, and the
        # two stream writes are not free (cfg2: 0.288 -> 0.301 ms)
        plans = getattr(self, 'render_plans', None)
        return bool(self.overlap and self._busy is not None and plans is not None and len(plans) > 1)

    def _set_busy(self, value):
        if self._use_busy():
            _lib.check(_lib.lib().ocrf_stream_write_value32(_lib.ptr(self._busy), int(value), _lib.stream_ptr(self.device)),
                       'ocrf_stream_write_value32')

    def _render_planned(self, entry, phase='both', out=None):
        plan, f0, nf, g = entry
        g = g['alt'][self.phase]
        bw = self.blend_workgroups
        # the "other chain still running" word: this path's own (several blends per step), or the one an owner that
        # renders this path on ITS side stream hands in (ShardedHotPath)
        word = getattr(self, '_yield_word', None)
        if word is None and self._use_busy():
            word = self._busy
        if bw == 'auto':
            # beside the other chain: 2.75 workgroups per CU of the five the chip holds (round 4: 3.5 with the heavier blend
            # behind a 32-us update; round 5, one host call per step and a blend of half the instructions: cfg2 step at
            # 448 / 512 / 576 / 640 / 704 / 768 / 896 workgroups 0.225 / 0.218 / 0.209 / 0.203 / 0.200 / 0.208 / 0.224 ms,
            # tools/sweep_r5_step.py)
            # Where the render outweighs the other chain by far (configs[4]: 721 k pixels per view against cfg2's 180 k, the
            # same poolings + HOA per frame) the optimum is 3.25 per CU: 768 / 800 / 832 / 864 / 896 / 960 workgroups
            # 1.806 / 1.783 / 1.777 / 1.787 / 1.820 / 1.889 ms against 1.879 at 704 (tools/sweep_cfg4_grid.py).
            H, W = self.cfg.input_size
            per_cu_x4 = 13 if H * W >= 512 * 1024 else 11
            bw = (per_cu_x4 * torch.cuda.get_device_properties(self.device).multi_processor_count // 4
                  if (self.overlap or getattr(self, '_yield_word', None) is not None) else 0)
        elif not isinstance(bw, int):
            bw = int(bw[min(f0, len(bw) - 1)])
        call_cams = None
        if self.plan_rebuild == 'per_step':
            if phase != 'blend':
                plan.rebuild(g['cams'])
            call_cams = g['cams']                       # ... and the call is checked against the plan's cameras on the device
        out = plan.render(g['rgb'], g['opacity'], g['scales'], g['rotations'], self.bg, guard=self.render_guard,
                          item_view=g['item_view'] if nf > 1 else None, blend_workgroups=bw, phase=phase, out=out,
                          yield_if=word if bw else None, cameras=call_cams, views_disjoint=True, want_radii=False)
        if phase == 'update':
            return out
        n = len(self.cams)
        return [{k: (v[i * n:(i + 1) * n] if k != 'status' else v) for k, v in out.items()} for i in range(nf)]

    def render(self, streams=None, want_n_contrib=False):
        """All owned cameras of every frame: list (one per frame) of dicts (``color``, ``depth``, ``final_T``, ...).
        ``streams``: a HIP stream per frame (frames on different streams get their own scratch buffer); the caller
        joins them.  The step is inference: the per-pixel contributor index (read only by the backward) is not
        tracked unless ``want_n_contrib`` (which renders per call, one call per frame)."""
        planned = self.render_mode == 'planned' and not want_n_contrib
        outs = []
        if planned:
            for entry in self._plans():       # built (one synchronisation each) before anything is enqueued
                if streams is None:
                    outs.extend(self._render_planned(entry))
                else:
                    with torch.cuda.stream(streams[entry[1]]):
                        outs.extend(self._render_planned(entry))
            return outs
        one_stream = streams is None or all(st is streams[0] for st in streams)
        if self.batch > 1 and one_stream and not want_n_contrib and self._per_call_sets() is not None:
            # every frame its own Gaussian set and cameras, all on one stream: ONE set of launches for the batch
            # (ocrf_rasterize_forward_sets: one preprocess / scan / scatter / blend over all frames' views — cfg2's two
            # frames 2 x 5 launches -> 5, and one blend of twelve views fills the chip better than two of six)
            g = self._per_call_sets()
            H, W = self.cfg.input_size
            ctx = torch.cuda.stream(streams[0]) if streams is not None else contextlib.nullcontext()
            with ctx:
                o = rasterize_sets(g['xyz'], g['rgb'], g['opacity'], g['scales'], g['rotations'], g['cams'], H, W, self.bg,
                                   workspace_tag='raster_sets')
            n = len(self.cams)
            return [{k: (v[b * n:(b + 1) * n] if k != 'status' else v) for k, v in o.items()} for b in range(self.batch)]
        for b in range(self.batch):
            if streams is None:
                outs.append(self._render_per_call(b, want_n_contrib, 'raster'))
            else:
                with torch.cuda.stream(streams[b]):
                    outs.append(self._render_per_call(b, want_n_contrib, f'raster{streams.index(streams[b])}'))
        return outs

    def _per_call_sets(self):
        """The frames' Gaussian sets stacked for one per-call render of the whole batch (built once; the per-frame
        tensors of ``frame_gauss`` stay the inputs of the per-frame path)."""
        if getattr(self, '_sets', None) is None:
            self._sets = False
            if self.batch * len(self.cams) <= 64:
                self._sets_by_phase = []
                for fgs in self.frame_gauss_sets:
                    g = {k: torch.stack([fg[k] for fg in fgs]).contiguous() for k in ('rgb', 'opacity', 'scales', 'rotations')}
                    if self._sets_by_phase:
                        g['xyz'], g['cams'] = self._sets_by_phase[0]['xyz'], self._sets_by_phase[0]['cams']
                    else:
                        g['xyz'] = torch.stack([self.voxel_xyz[b].reshape(-1, 3) for b in range(self.batch)]).contiguous()
                        g['cams'] = torch.cat([rc['packed'] for rc in self.frame_cams]).contiguous()
                    self._sets_by_phase.append(g)
                self._sets = self._sets_by_phase[self.phase]
        return self._sets or None

    def set_phase(self, phase):
        """Switch to parameter set ``phase`` of every frame (``alternate``): the next step renders (and HOA reads) the
        other tensors — new parameters, as a network's heads hand them over every step.  A recorded step is per phase."""
        phase = int(phase) % len(self.frame_gauss_sets)
        if phase == self.phase:
            return
        self.phase = phase
        self.frame_gauss = self.frame_gauss_sets[phase]
        self.gauss = self.frame_gauss[0]
        if getattr(self, '_opac_flat', None) is not None:
            self._opac_flat = self._opac_flats[phase]
        if getattr(self, '_sets', None):
            self._sets = self._sets_by_phase[phase]

    @property
    def views_per_step(self):
        return self.batch * len(self.cams) if self.cfg.render else 0

    def _or_empty(self, five):
        if five[0] is None:
            e = torch.zeros(0, dtype=torch.int32, device=self.device)
            return e, e, e, e, e
        return five

    def make_inputs(self, seed=0):
        """depth (B, N, D, H, W), feat (B, N, H, W, C) channels-last, on the device."""
        cfg = self.cfg
        full = synthetic.PathConfig(**{**cfg.__dict__, 'batch': self.batch})
        depth, feat = synthetic.depth_and_feat(full, seed)
        Hf, Wf = cfg.feat_hw
        depth = depth.view(self.batch, cfg.n_cams, cfg.D, Hf, Wf)[:, self.cams].contiguous()
        feat = feat.view(self.batch, cfg.n_cams, cfg.channels, Hf, Wf)[:, self.cams]
        feat = feat.permute(0, 1, 3, 4, 2).contiguous()
        return depth.to(self.device), feat.to(self.device)

    def _panel_plan(self, plan):
        if plan.mfma_plan is None:
            backend = self.ht_pool_backend if plan is self.ht else self.lss_pool_backend
            if plan is self.ht:
                kw = dict(group=8)
            elif backend == 'panel':
                kw = dict(group=8, unit_cost=self.lss_panel_unit_cost)
            else:
                kw = dict(group=self.lss_mfma_group)
            plan.mfma_plan = bevpool.MfmaPoolPlan(plan.ranks_depth, plan.ranks_feat, plan.ranks_bev, plan.bev_shape, **kw)
        return plan.mfma_plan

    def pool(self, plan, depth, feat, out=None, weights_ready=False):
        """-> (B, Z*C, Y, X): pooled BEV with Z collapsed into channels (view_transformer.py:194).  The
        rank vectors are cached, so the rank-only half of the pooling is too (bevpool.DevicePoolPlan).
        ``out``: contiguous fp32 tensor to write into (a slice of a fused buffer)."""
        if plan.n_points == 0:
            res = bevpool.bev_pool_v2_collapsed(depth, feat, plan.ranks_depth, plan.ranks_feat,
                                                plan.ranks_bev, plan.bev_shape, plan.starts, plan.lengths)
            if out is not None:
                out.view_as(res).copy_(res)
            return res
        backend = self.ht_pool_backend if plan is self.ht else (self.lss_pool_backend if plan is self.lss else 'tile')
        if backend in ('mfma', 'panel') and plan.bev_shape[-1] in (64, 80, 96, 128):
            # the panel plan (unique feature rows of an 8 x 8 tile in panels of 48; cells = (voxel, row) pairs).  'panel':
            # cells walked out of LDS after a weight pre-pass (csrc/bev_pool_panel.hip); 'mfma': dense W . F panels on the
            # matrix cores (csrc/bev_pool_mfma.hip).  The height-sampling ranks have no heavy tiles (11 points per feature
            # row and 8x8 tile): units of up to 8 panels; the LSS ranks do (the block beside the rig): units cut by
            # estimated cost ('panel') / short units ('mfma'), more slabs
            self._panel_plan(plan)
            if backend == 'mfma':
                return bevpool.bev_pool_v2_mfma(depth, feat, plan.mfma_plan, out=out)
            return bevpool.bev_pool_v2_panel(depth, feat, plan.mfma_plan, out=out, weights_ready=weights_ready)
        if plan.device_plan is None:
            plan.device_plan = bevpool.DevicePoolPlan(plan.ranks_depth, plan.ranks_feat, plan.ranks_bev, plan.bev_shape,
                                                      plan.starts, plan.lengths)
        return bevpool.bev_pool_v2_planned(depth, feat, plan.device_plan, out=out)

    def _camera_blocks(self):
        cfg, dev, args = self.cfg, self.device, self._calib_host
        if self.device_geometry and hasattr(self, '_calib_dev'):
            lss_block, ht_block, _ = index_prep.geometry_blocks_hip(*self._calib_dev, None, cfg.input_size)
        else:
            lss_block = index_prep.lss_camera_block(*args).to(dev, non_blocking=True)
            lidar2img, img_aug, _, _ = index_prep.get_projection(*args)
            ht_block = index_prep.ht_camera_block(lidar2img, img_aug).to(dev, non_blocking=True)
        return lss_block, ht_block

    def _prepare_lss(self, lss_block):
        B, N = self._calib_host[1].shape[:2]
        return index_prep.voxel_pooling_prepare_v2_hip(self._frustum_dev, lss_block, B, N, *self._grid,
                                                       buffers=self._lss_bufs, sync=False)

    def _prepare_ht(self, ht_block):
        cfg = self.cfg
        B, N = self._calib_host[1].shape[:2]
        Hf, Wf = cfg.feat_hw
        return index_prep.fast_sample_prepare_hip(self._ref_template, ht_block, B, N, list(cfg.pc_range), cfg.input_size,
                                                  cfg.grid['depth'], Wf, Hf, cfg.D, buffers=self._ht_bufs, sync=False)

    def prepare_indices_hip(self, sync=True):
        """Rank vectors of both poolings from the calibration, on the device (view_transformer.py:108-147,
        197-255; view_transformer_ocrf.py:675-740,785-852): tiny per-camera algebra on the host, one
        small upload each, then csrc/index_prep.hip.  -> (lss PoolPlan, ht PoolPlan)."""
        cfg, dev = self.cfg, self.device
        X, Y, Z = cfg.bev_xyz
        Hf, Wf = cfg.feat_hw
        args = self._calib_host
        B, N = args[1].shape[:2]
        lss_block, ht_block = self._camera_blocks()
        lss = self._prepare_lss(lss_block)
        ht = self._prepare_ht(ht_block)
        if not sync:
            return lss, ht                                      # ((five capacity vectors), counts) each
        counts = torch.stack((lss[1], ht[1])).cpu()           # ONE device->host read for both (4 ints)
        return (PoolPlan(*self._or_empty(index_prep._trim(lss[0], counts[0])), (self.batch, Z, Y, X, cfg.channels)),
                PoolPlan(*self._or_empty(index_prep._trim(ht[0], counts[1])), (self.batch, 1, Y, X, cfg.channels)))

    def pool_step(self, depth, feat, prepared=None):
        """Both poolings (index preparation first in 'per_step' mode): LSS BEV (B, Z*C, Y, X) and
        HT BEV (B, C, Y, X) (view_transformer.py:194, view_transformer_ocrf.py:781).  ``prepared``: what
        ``prepare_indices_hip(sync=False)`` returned, if the caller already issued it."""
        if self.index_prep_mode == 'per_step':
            # ranks stay on the device, their lengths too: no host read anywhere in the step
            (lv, lc), (hv, hc) = prepared if prepared is not None else self.prepare_indices_hip(sync=False)
            lss = bevpool.bev_pool_v2_device_counts(depth, feat, lv[1], lv[2], lv[0], self.lss.bev_shape, lv[3], lv[4], lc)
            ht = bevpool.bev_pool_v2_device_counts(depth, feat, hv[1], hv[2], hv[0], self.ht.bev_shape, hv[3], hv[4], hc)
            return lss, ht
        C = self.lss.bev_shape[-1]
        if (self.lss_pool_backend == 'panel' and self.ht_pool_backend == 'panel' and C in (64, 80, 96, 128)
                and self.lss.n_points and self.ht.n_points):
            # both poolings read the same depth tensor: ONE launch sums the cell weights of both plans
            bevpool.bev_pool_cell_weights(depth, self._panel_plan(self.lss), self._panel_plan(self.ht))
            return self.pool(self.lss, depth, feat, weights_ready=True), self.pool(self.ht, depth, feat, weights_ready=True)
        return self.pool(self.lss, depth, feat), self.pool(self.ht, depth, feat)

    def step(self, depth, feat):
        """One pass of the hot path: pools (+ render + HOA where the configuration has them).
        -> (lss, ht[, rendered][, gated, opacity_bev]), everything ordered on the caller's stream.  With ``one_call``
        the returned tensors are the same objects every step (see ``__init__``)."""
        if self.cfg.render and self.render_mode == 'planned':
            self._plans()             # first step: built on the caller's stream BEFORE the side streams branch off it
        if not (self.one_call and self._one_call_ok) or torch.cuda.is_current_stream_capturing():
            return self._step_eager(depth, feat)
        cur = torch.cuda.current_stream(self.device)
        key = self._step_key(depth, feat)
        hit = self._compiled.get(key)
        if hit is not None and hit[3] != _lib.workspace.generation(self.device):
            # a scratch buffer the recording baked in was replaced since (a larger request for its tag, by anyone in the
            # process): its pointers are stale — record anew
            del self._compiled[key]
            hit = None
        if hit is None:
            if key not in self._warm_keys:
                # the first step of a (tensors, mode) builds plans and scratch (launches that belong to no later step):
                # issued call by call, the next one is recorded
                if len(self._warm_keys) >= 16:
                    self._warm_keys.clear()
                self._warm_keys.add(key)
                return self._step_eager(depth, feat)
            return self._record_step(depth, feat, key, cur)
        compiled, out, _pool, _gen = hit
        compiled.run(*[st.cuda_stream for st in self._step_streams(cur)])
        for entry in (self.render_plans or []) if self.cfg.render and self.render_mode == 'planned' else []:
            entry[0]._host_guarded = entry[0]._host_guarded or self.render_guard == 'host'
        return out

    def _step_key(self, depth, feat):
        """Everything ``_step_eager`` reads that decides WHICH launches a step is (a recorded step must not
        outlive a changed mode attribute): the input tensors (address, shape, dtype), the parameter phase and the modes."""
        bw = self.blend_workgroups
        return (depth.data_ptr(), feat.data_ptr(), tuple(depth.shape), tuple(feat.shape), depth.dtype, feat.dtype, self.phase,
                self.index_prep_mode, self.plan_rebuild, self.overlap, bw if isinstance(bw, (int, str)) else tuple(bw),
                self.render_mode, self.render_guard, self.hoa_first, self.lss_pool_backend, self.ht_pool_backend,
                self.render_streams, self.fuse_frames, self.device_geometry)

    def _forks(self):
        return self.overlap and self.cfg.render

    def _prep_forks(self):
        """The per-step index preparation on two streams of its own (``_main_chain``)."""
        return (self.index_prep_mode == 'per_step' and self.cfg.hoa and self.overlap and self.device_geometry
                and hasattr(self, '_calib_dev'))

    def _step_streams(self, cur):
        """The step's streams in slot order: the caller's, the render streams, the two index-preparation streams."""
        st = [cur]
        if self._forks():
            if not self._side:
                self._side = [shared_stream(self.device, 'render' if k == 0 else f'render{k}')
                              for k in range(self.render_streams)]
            st += list(self._side)
        if self._prep_forks():
            if self._prep_stream is None:
                self._prep_stream = shared_stream(self.device, 'prep')
                self._prep_stream2 = shared_stream(self.device, 'prep2')
            st += [self._prep_stream, self._prep_stream2]
        return st

    def _record_step(self, depth, feat, key, cur):
        """Issue the step eagerly once more, with the library calls logged (``_lib.StepRecorder``) and every tensor it
        allocates drawn from a memory pool of its own (the recorded pointers stay valid: the pool is kept)."""
        rec = _lib.StepRecorder(self._step_streams(cur))
        pool = torch.cuda.MemPool()
        with torch.cuda.use_mem_pool(pool, self.device), rec:
            out = self._step_eager(depth, feat, rec)
        if not rec.ok or (self.index_prep_mode == 'per_step' and not (self.device_geometry and hasattr(self, '_calib_dev'))):
            # a step with launches outside the recordable entry points — or the per-step index preparation with the
            # calibration algebra on the HOST (torch CPU ops + uploads per step, which a replay would not repeat): keep
            # issuing it call by call
            self._one_call_ok = False
            self.one_call_refused = rec.why or 'per-step calibration algebra on the host is issued call by call'
            return out
        compiled = rec.build()
        # The recorder sees the library's calls only: a torch kernel issued inside the step (a ``zero_``, a ``copy_``, a
        # hidden ``.contiguous()``) would be missing from the replay, silently.  So the FIRST replay is held
        # to the step that was just issued call by call, bit for bit, on the same inputs — one synchronisation, at record
        # time only; a recording that does not reproduce itself is dropped and the step stays call by call.
        flat = [t for t in self._flat_outputs(out)]
        want = [t.clone() for t in flat]
        compiled.run(*[st.cuda_stream for st in self._step_streams(cur)])
        torch.cuda.synchronize(self.device)
        if not all(torch.equal(a, b) for a, b in zip(flat, want)):
            self._one_call_ok = False
            self.one_call_refused = 'the first replay of the recorded step did not reproduce the step issued call by call'
            return self._step_eager(depth, feat)
        while len(self._compiled) >= 4:                    # (two phases x two input pairs, with room)
            self._compiled.pop(next(iter(self._compiled)))
        self._compiled[key] = (compiled, out, pool, _lib.workspace.generation(self.device))
        return out

    @staticmethod
    def _flat_outputs(out):
        for o in out:
            if torch.is_tensor(o):
                yield o
            elif isinstance(o, dict):
                yield from (v for k, v in o.items() if torch.is_tensor(v) and k != 'status')
            elif isinstance(o, (list, tuple)):
                yield from HotPath._flat_outputs(o)

    def _step_eager(self, depth, feat, rec=None):
        fork = self._forks()
        self._rec = rec
        if not fork:
            main = self._main_chain(depth, feat)
            rendered = [self.render()] if self.cfg.render else []
            return tuple(main[:2]) + tuple(rendered) + tuple(main[2:])
        cur = torch.cuda.current_stream(self.device)
        # "the main chain is running": the persistent blends of the render stream keep to 2.75 - 3.25 workgroups per CU
        # while it is up and take the whole chip once it is down (cfg4: the renders outlast the poolings + HOA)
        self._set_busy(1)
        if not self._side:
            self._side = [shared_stream(self.device, 'render' if k == 0 else f'render{k}')
                          for k in range(self.render_streams)]
        # (wait_stream = create an event + record + wait: the events are kept and reused, ~5 us of host time per call)
        if getattr(self, '_fork_ev', None) is None:
            self._fork_ev = torch.cuda.Event()
            self._join_ev = [torch.cuda.Event() for _ in self._side]
        self._fork_ev.record(cur)
        for k, side in enumerate(self._side):
            side.wait_event(self._fork_ev)        # inputs (and last step's consumers) are ordered before
            if rec is not None:
                rec.fork(0, 1 + k)
        # host issue order: the render call first (measured against the LSS pooling / both poolings first: within 1 %)
        rendered = self.render([self._side[b % len(self._side)] for b in range(self.batch)])
        main = self._main_chain(depth, feat)
        self._set_busy(0)
        for k, (side, ev) in enumerate(zip(self._side, self._join_ev)):
            ev.record(side)
            cur.wait_event(ev)                    # join: everything the step returns is ordered on `cur`
            if rec is not None:
                rec.join(1 + k, 0)
        # the rendered images were allocated while a side stream was current and are consumed on the caller's: the caching
        # allocator must not hand their blocks to the next side-stream render while the caller still reads them
        for frame in rendered:
            for v in frame.values():
                if torch.is_tensor(v):
                    v.record_stream(cur)
        return tuple(main[:2]) + (rendered,) + tuple(main[2:])

    def _main_chain(self, depth, feat):
        """Pools + HOA on the current stream -> (lss, ht[, gated, opacity_bev])."""
        # HOA-1/2 do not read the pooled BEV and are latency chains of small kernels
        prepared = None
        if self._prep_forks():
            # the index preparation (~ 20 launches, 0.17 ms) needs nothing HOA-1/2 produce: on streams of its own
            # beside them; the poolings wait for it.  Only with the calibration algebra on the device: with the host
            # formulation the step is bound by the host (its ~ 20 small CPU torch ops + ~ 60 launches), and the extra
            # stream / event calls cost more than the overlap returns (measured with pinned upload slots: 0.55 -> 0.61-0.70 ms)
            # The two preparations are independent chains of ~ 10 / 5 launches (89 / 62 us alone): one stream each
            # (their look-back scratch is per stream, index_prep._prep_tag), the calibration blocks computed once on
            # the first and handed over by an event; each pooling waits only for its own ranks.
            cur0 = torch.cuda.current_stream(self.device)
            self._step_streams(cur0)                                 # (creates the two streams on first use)
            p1, p2 = self._prep_stream, self._prep_stream2
            rec = getattr(self, '_rec', None)
            s1 = 1 + (len(self._side) if self._forks() else 0)      # stream slots of p1, p2 (``_step_streams``)
            p1.wait_stream(cur0)
            p2.wait_stream(cur0)
            if rec is not None:
                rec.fork(0, s1)
                rec.fork(0, s1 + 1)
            # Issue order = the critical chain first (round 6, from the step's kernel timeline, tools/trace_mode.sh): the LSS
            # preparation -> LSS pooling -> HOA-3 chain is the longest, and a step's ~ 45 launches cost the host ~ 4 us
            # each.  So: LSS preparation AND its pooling on their stream, then the HT pair on theirs, then HOA-1/2 on the
            # caller's (it only has to be done before HOA-3).  Round 5 pooled the LSS ranks on the caller's stream behind
            # HOA-1/2: in the timeline the pooling started ~ 40 us after its ranks were ready.  Measured, one box: 0.375 ->
            # 0.381 ms (per_step_devgeom 0.332 -> 0.338): neutral — beside the render's blend the step is bound by the
            # chip's occupancy, not by this chain; kept because both poolings are now issued the same way.
            with torch.cuda.stream(p1):
                lss_block, ht_block = self._camera_blocks()
                blocks_ready = torch.cuda.Event()
                blocks_ready.record(p1)
                ht_block.record_stream(p2)
                if rec is not None:
                    rec.fork(s1, s1 + 1)                             # (recorded here: the event is taken before the LSS chain)
                lv, lc = self._prepare_lss(lss_block)
                lss = bevpool.bev_pool_v2_device_counts(depth, feat, lv[1], lv[2], lv[0], self.lss.bev_shape, lv[3], lv[4], lc)
            p2.wait_event(blocks_ready)
            with torch.cuda.stream(p2):
                hv, hc = self._prepare_ht(ht_block)
                ht = bevpool.bev_pool_v2_device_counts(depth, feat, hv[1], hv[2], hv[0], self.ht.bev_shape, hv[3], hv[4], hc,
                                                       scratch_tag='bev_pool_nchw_b')
            prepared = True
        # HOA-1/2 need nothing of the poolings and the poolings nothing of them: with the planned render the poolings go
        # FIRST — they meet the start of the side stream's blend instead of its middle; with the per-call render, whose
        # chip-filling preprocess opens the side stream, and with the per-step index preparation (which the poolings have
        # to wait for anyway) HOA-1/2 first.
        hoa_first = self.hoa_first
        if hoa_first is None:
            hoa_first = self.render_mode != 'planned' or self.index_prep_mode == 'per_step'
        ob = self.hoa_opacity_bev() if (self.cfg.hoa and hoa_first) else None
        if prepared is not None:
            main = torch.cuda.current_stream(self.device)
            rec = getattr(self, '_rec', None)
            s1 = 1 + (len(self._side) if self._forks() else 0)
            main.wait_stream(self._prep_stream)
            if rec is not None:
                rec.join(s1, 0)
            main.wait_stream(self._prep_stream2)
            if rec is not None:
                rec.join(s1 + 1, 0)
            lss.record_stream(main)
            ht.record_stream(main)
        else:
            lss, ht = self.pool_step(depth, feat, prepared)
        if self.cfg.hoa and not hoa_first:
            ob = self.hoa_opacity_bev()
        out = [lss, ht]
        if self.cfg.hoa:
            # stand-in for geom_feat: the HT BEV has its shape (B,C,Y,X); the fusion convs between
            # the pools and HOA-3 are MIOpen territory, not part of this path
            out.extend(self.hoa_step(ht, ob))
        return out

    @property
    def bev_voxels_per_step(self):
        X, Y, Z = self.cfg.bev_xyz
        return self.batch * Z * Y * X


class ShardedHotPath:
    """The hot path of ONE sample with its camera-frames sharded over the ranks of a ``torch.distributed`` job
    (``sharding.CameraFramePlan``; BASELINE.json north_star, configs[3] / [4]): this rank pools and renders only
    the camera-frames it owns and runs HOA only for the frames it has a part in.  Per step:
      renders (side HIP stream) | pools of the owned cameras into the fused grid's plane blocks
      step 1 of the exchange: the partial grids are summed inside each frame's group (reduce_scatter over plane blocks)
      HOA-1/2 of the OWN frames (they read nothing of the pooled BEV: issued beside step 1)
      HOA-3 on the rank's finished plane blocks, in place (``sharding.gate_blocks``: per-block channel statistics, one
        small all_gather inside the group, the gate of the own channels) — the opacity BEV goes into an extra plane
      step 2: ONE world all_gather — it carries the LSS planes, the GATED height-sampling planes and the opacity BEVs:
        the complete fused grid ``(n_frames, Z*C + C + 1, Y, X)`` on every rank.
    HOA is a per-sample loop in the reference (view_transformer_ocrf.py:1090-1161,1196-1199) and frames are independent
    until the concat (detectors/ocrfdet.py:274): at configs[4] on 8 ranks (whole frames per rank, no reduce) a rank runs the
    ten HOA launches of ONE frame, not of eight; with groups, HOA-1/2 are replicated inside a frame's group only and
    HOA-3 is split over it.  The collectives are asynchronous: the renders and HOA-1/2 run beside them.  With
    ``world == 1`` it is ``HotPath`` with the same fused output buffer, bit for bit."""

    def __init__(self, cfg, device, rank, world, index_prep_mode='cached', render_mode='planned', render_guard='host',
                 sparse_exchange=True, collectives=None, one_call=True):
        from . import sharding
        self.cfg, self.device, self.rank, self.world = cfg, torch.device(device), rank, world
        # one_call: the rank's COMPUTE between the collectives — (poolings + renders) and (HOA-1/2) — is recorded once and
        # replayed by one host call each (``_lib.StepRecorder``, as ``HotPath.step``): a rank of eight has ~ 0.1 ms of
        # device work per step and would otherwise spend ~ 0.2 ms issuing its ~ 16 library calls through ctypes
        self.one_call = bool(one_call) and self.device.type == 'cuda' and index_prep_mode == 'cached'
        self._segments, self._segment_seen, self.one_call_refused = {}, {}, None
        X, Y, Z = cfg.bev_xyz
        C = cfg.channels
        self.n_frames = cfg.batch * cfg.n_frames
        self.planes_lss = Z * C
        self.planes_pool = (Z + 1) * C                          # LSS planes, then HT planes: what the poolings write
        # the frame's opacity BEV rides along as one more plane of the fused grid (written by the rank whose block holds it)
        self.opacity_plane = self.planes_pool if cfg.hoa else None
        self.plan = sharding.CameraFramePlan(cfg.n_cams, self.n_frames, world, self.planes_pool + (1 if cfg.hoa else 0))
        self.exchange = sharding.BevExchange(self.plan, rank, self.device, (Y, X), collectives=collectives)
        one = synthetic.PathConfig(**{**cfg.__dict__, 'batch': 1, 'n_frames': 1, 'hoa': False})
        self.subs = {f: HotPath(one, self.device, cams=self.plan.cams_of(rank, f), index_prep_mode=index_prep_mode,
                                overlap=False, frame_offset=f, render_mode=render_mode, render_guard=render_guard,
                                one_call=False)
                     for f in self.plan.frames_of(rank)}
        # HOA weights and inputs: drawn for ALL frames exactly as the unsharded HotPath draws them (one generator), then
        # only the own frames' slices are kept on the device for the steps
        self.my_frames = self.plan.frames_of(rank)
        self.base = None
        if cfg.hoa:
            self.base = HotPath(synthetic.PathConfig(**{**cfg.__dict__, 'render': cfg.render or cfg.hoa}), self.device,
                                cams=[0], index_prep_mode='cached', overlap=False, one_call=False)
            fr = self.my_frames
            Zh = cfg.num_height
            self._hoa_in = None
            if fr:
                opac = torch.stack([self.base.frame_gauss[f]['opacity'].view(Zh, Y, X) for f in fr]).reshape(-1, 1).contiguous()
                self._hoa_in = (opac, self.base.alpha_lidar[fr].contiguous(), self.base.bev_pos1[fr].contiguous())
        self.hoa_launch_frames = 0                              # frames whose HOA-1/2 this rank issued in its last step
        self._side = shared_stream(self.device, 'render') if self.device.type == 'cuda' and cfg.render else None
        # the renders run on the side stream beside this rank's poolings, the exchange and HOA: their persistent blends
        # keep to a part of the chip while that chain is running and take the whole chip once it is done (the same
        # occupancy split as HotPath.step; a hint, the images do not depend on it)
        self._busy = None
        if self._side is not None and render_mode == 'planned':
            self._busy = torch.zeros(1, dtype=torch.int32, device=self.device)
            for sub in self.subs.values():
                sub._yield_word = self._busy
        # wedge-sparse step 1: a member's partial grid is zero outside the strips its cameras' rank vectors touch
        # (static per calibration) — only those strips of the other members' plane blocks cross xGMI
        # The decision is a function of the PLAN (the same on every rank), never of this rank's own buffers: set_touched
        # is a world collective, so idle ranks and owners of whole frames take part with an empty dict.  Only with cached
        # rank vectors: with per-step index preparation (the reference's accelerate=False, calibration per sample) a new
        # calibration may touch tiles outside a stale list and its contributions would be dropped from the sum.
        self.sparse_exchange = bool(sparse_exchange) and self.plan.any_shared and index_prep_mode == 'cached'
        if self.sparse_exchange:
            ex, touched = self.exchange, {}
            for f, sub in self.subs.items():
                if f in ex.partial:
                    yx = torch.cat((sub.lss.ranks_bev.long() % (Y * X), sub.ht.ranks_bev.long() % (Y * X)))
                    touched[f] = torch.unique(ex.tile_of_voxel(yx))
            ex.set_touched(touched)

    def make_inputs(self, seed=0):
        """Per owned frame: depth (1, n_owned_cams, D, H, W), feat (1, n_owned_cams, H, W, C) — the same values
        the unsharded ``HotPath.make_inputs(seed)`` holds for those camera-frames."""
        cfg = self.cfg
        full = synthetic.PathConfig(**{**cfg.__dict__, 'batch': self.n_frames})
        depth, feat = synthetic.depth_and_feat(full, seed)
        Hf, Wf = cfg.feat_hw
        depth = depth.view(self.n_frames, cfg.n_cams, cfg.D, Hf, Wf)
        feat = feat.view(self.n_frames, cfg.n_cams, cfg.channels, Hf, Wf).permute(0, 1, 3, 4, 2)
        out = {}
        for f, sub in self.subs.items():
            out[f] = (depth[f:f + 1, sub.cams].contiguous().to(self.device), feat[f:f + 1, sub.cams].contiguous().to(self.device))
        return out

    @property
    def views_per_step(self):
        return sum(len(s.cams) for s in self.subs.values()) if self.cfg.render else 0

    # ---- HOA, sharded by frame ----------------------------------------------------------------------------------
    @torch.no_grad()
    def _hoa12(self, slot=0):
        """HOA-1/2 of the frames this rank has a part in -> {frame: (Y,X) opacity BEV} (empty on an idle rank).
        ``slot``: which output buffer set a REPLAYED segment writes — the pipelined step alternates two, because step k's
        HOA-3 gate reads these tensors on the communication stream while step k + 1's replay would already rewrite them on
        the caller's (a write-after-read race that constant test inputs hid)."""
        self.hoa_launch_frames = 0
        if self.base is None or not self.my_frames:
            return {}
        cfg = self.cfg
        X, Y, _ = cfg.bev_xyz
        opac, alpha, pos = self._hoa_in
        m = self.base.hoa_mods

        def body(rec):
            return m['v2b'](hoa.hoa1(m['dca'], opac, alpha, cfg.num_height, Y, X), pos)      # (len(my_frames), 1, Y, X)
        if self.device.type == 'cuda':
            ob = self._segment('hoa12', (opac.data_ptr(), alpha.data_ptr(), pos.data_ptr(), int(slot)),
                               [torch.cuda.current_stream(self.device)], body)
        else:
            ob = body(None)
        self.hoa_launch_frames = len(self.my_frames)
        return {f: ob[i, 0] for i, f in enumerate(self.my_frames)}

    def _gate(self, ex, ob):
        """HOA-3 on this rank's finished plane blocks, in place (``sharding.gate_blocks``)."""
        from . import sharding
        if self.base is None:
            return
        X, Y, _ = self.cfg.bev_xyz
        mask = self.base.hoa_mods['mask']

        def stats_fn(x):
            return hoa.channel_stats(x.unsqueeze(0))[0]

        def gate_fn(x, stats, opacity):
            mask.gate(x.unsqueeze(0), opacity.reshape(1, 1, Y, X), stats=stats.unsqueeze(0).contiguous(), in_place=True)
        def body(rec):
            with torch.no_grad():
                sharding.gate_blocks(ex, self.planes_lss, self.cfg.channels, None, ob, stats_fn, gate_fn,
                                     partial_ok=getattr(self, 'partial_statistics_ok', False))
        # a rank whose frames are all its own (no group to gather statistics from) gates without a collective inside:
        # one more recorded segment
        alone = all(len(self.plan.group_of_frame[f]) == 1 or not ex.active for f, _, _ in ex.my_blocks)
        if alone and self.device.type == 'cuda':
            key = (id(ex),) + tuple(t.data_ptr() for t in ob.values())
            self._segment('gate', key, [torch.cuda.current_stream(self.device)], body)
        else:
            body(None)
        # the frame's opacity BEV into its plane of the fused grid (the member whose block holds that plane): it travels
        # with the gather (a torch copy: not part of a recorded segment)
        with torch.no_grad():
            for f, p0, n, view in ex.block_views():
                if p0 <= self.opacity_plane < p0 + n:
                    view[self.opacity_plane - p0].copy_(ob[f])

    def _outputs(self, full, rendered):
        gated = opacity_bev = None
        if self.base is not None:
            gated = [full[f:f + 1, self.planes_lss:self.planes_pool] for f in range(self.n_frames)]
            opacity_bev = full[:, self.planes_pool:self.planes_pool + 1]
        return full, rendered, gated, opacity_bev

    def _segment(self, name, key, streams, fn):
        """``fn(rec)`` — library calls only, on ``streams`` (slot 0 = the caller's) — by ONE host call from the third time
        ``key`` (the pointers it works on) is seen: first eagerly (plans and scratch are built), then recorded with its
        allocations in a memory pool of its own, then replayed.  -> what ``fn`` returned (the same tensors every replay)."""
        if not self.one_call or torch.cuda.is_current_stream_capturing():
            return fn(None)
        hit = self._segments.get((name, key))
        if hit is not None and hit[3] != _lib.workspace.generation(self.device):
            del self._segments[(name, key)]          # a scratch buffer it baked in was replaced: record anew
            hit = None
        if hit is not None:
            hit[0].run(*[st.cuda_stream for st in streams])
            return hit[1]
        if (name, key) not in self._segment_seen:
            # (bounded: a caller that hands over NEW tensors every step never repeats a key — it runs call by call, and
            # these tables must not grow with it)
            if len(self._segment_seen) >= 32:
                self._segment_seen.clear()
            self._segment_seen[(name, key)] = True
            return fn(None)
        rec, pool = _lib.StepRecorder(streams), torch.cuda.MemPool()
        with torch.cuda.use_mem_pool(pool, self.device), rec:
            out = fn(rec)
        if not rec.ok:
            self.one_call, self.one_call_refused = False, rec.why
            return out
        while len(self._segments) >= 16:                    # (four segments x two buffer sets, with room)
            self._segments.pop(next(iter(self._segments)))
        self._segments[(name, key)] = (rec.build(), out, pool, _lib.workspace.generation(self.device))
        return out

    def _pool_and_render(self, inputs, target_of):
        cur = torch.cuda.current_stream(self.device) if self._side is not None else None
        targets = {f: target_of(f) for f in self.subs}

        def body(rec):
            rendered = []
            if self._side is not None:
                self._set_busy(1)
                self._side.wait_stream(cur)
                if rec is not None:
                    rec.fork(0, 1)
                for f, sub in self.subs.items():
                    rendered.append(sub.render([self._side]))
            for f, sub in self.subs.items():
                depth, feat = inputs[f]
                tgt = targets[f]
                sub.pool(sub.lss, depth, feat, out=tgt[:self.planes_lss])
                sub.pool(sub.ht, depth, feat, out=tgt[self.planes_lss:self.planes_pool])
            return rendered
        if self.device.type != 'cuda' or not self.subs:
            return cur, body(None)
        key = tuple((f, inputs[f][0].data_ptr(), inputs[f][1].data_ptr(), targets[f].data_ptr()) for f in self.subs)
        main = cur if cur is not None else torch.cuda.current_stream(self.device)
        streams = [main] + ([self._side] if self._side is not None else [])
        return cur, self._segment('pool_render', key, streams, body)

    def _join_renders(self, cur, rendered):
        if self._side is not None:
            self._set_busy(0)
            cur.wait_stream(self._side)
            for per_sub in rendered:          # allocated while the side stream was current, consumed on the caller's
                for frame in per_sub:
                    for v in frame.values():
                        if torch.is_tensor(v):
                            v.record_stream(cur)

    def step(self, inputs):
        """-> (fused BEV (n_frames, Z*C + C [+ 1], Y, X) complete on every rank — LSS planes, GATED height-sampling planes,
        opacity BEV —, rendered list, gated (views of the fused grid, per frame), opacity_bev (view))."""
        if getattr(self, 'pipe', None) is not None and self.pipe._pending is not None:
            raise _lib.OcrfHipError('a pipelined step is pending on the shared exchange buffers: flush_pipelined() first')
        ex = self.exchange
        cur, rendered = self._pool_and_render(inputs, ex.pool_target)
        works = ex.start()
        ob = self._hoa12()
        ex.finish_reduce(works)
        self._gate(ex, ob)
        full = ex.gather()
        self._join_renders(cur, rendered)
        return self._outputs(full, rendered)

    def step_pipelined(self, inputs):
        """The same step with the exchange taken off its critical path (``sharding.PipelinedExchange``): this call pools
        and renders step k, runs its HOA-1/2 and starts its exchange — step 1, the in-place HOA-3 of the own blocks, step 2
        — on a communication stream; it RETURNS step k - 1 (as ``step`` does; ``None`` for the first call), whose exchange
        ran under this call's poolings and renders.  ``flush_pipelined()`` hands out the last step.  A rank's throughput
        is then max(own compute, exchange) instead of their sum; the price is one step of latency, and the returned
        tensors of step k - 1 are overwritten two calls later.  The pipeline has two buffer sets of its own on the process
        groups of ``self.exchange``."""
        from . import sharding
        if getattr(self, 'pipe', None) is None:
            ex0 = sharding.BevExchange(self.plan, self.rank, self.device, (self.exchange.Y, self.exchange.X),
                                       share=self.exchange)
            if self.exchange.touched:
                ex0.adopt_touched(self.exchange)
            self.pipe = sharding.PipelinedExchange(self.plan, self.rank, self.device, (self.exchange.Y, self.exchange.X),
                                                   first=ex0)
            self._held = None
        pipe = self.pipe
        cur, rendered = self._pool_and_render(inputs, pipe.pool_target)
        self._pipe_k = getattr(self, '_pipe_k', 0) + 1
        ob = self._hoa12(slot=self._pipe_k % 2)
        if pipe._comm is not None:
            for t in ob.values():
                t.record_stream(pipe._comm)
        full_prev = pipe.submit(between=lambda ex: self._gate(ex, ob))      # step k's exchange starts; step k - 1's grid is complete
        held, self._held = self._held, rendered
        self._join_renders(cur, rendered)
        if full_prev is None or held is None:
            return None
        return self._outputs(full_prev, held)

    def flush_pipelined(self):
        """-> the last step submitted by ``step_pipelined`` (None if nothing is pending)."""
        if getattr(self, 'pipe', None) is None:
            return None
        held, self._held = self._held, None
        full = self.pipe.flush()
        if full is None or held is None:
            return None
        return self._outputs(full, held)

    def _set_busy(self, value):
        if self._busy is not None:
            _lib.check(_lib.lib().ocrf_stream_write_value32(_lib.ptr(self._busy), int(value), _lib.stream_ptr(self.device)),
                       'ocrf_stream_write_value32')

    @property
    def bev_voxels_per_step(self):
        X, Y, Z = self.cfg.bev_xyz
        return self.n_frames * Z * Y * X


class NeckPath:
    """The whole neck as one step: pre-filter + ``OcRFViewTransformerFull.view_transform`` (both
    poolings, colour / alpha sampling, Gaussian heads, NeRF branch, one rendered view per sample, HOA,
    BEV fusion) with random-init weights of the reference architecture on the synthetic rig — what
    ``bench.py --scope neck`` and ``tools/time_neck.py`` drive.  Frames ride along as batch entries."""

    def __init__(self, cfg, device, accelerate=True, seed=0, parallel_branches=True, host_calibration=False):
        from . import neck_ops
        from . import view_transformer_ocrf as vto
        self.cfg, self.device, self._ops = cfg, torch.device(device), neck_ops
        self.batch = cfg.batch * cfg.n_frames
        X, Y, _ = cfg.bev_xyz
        torch.manual_seed(seed)
        self.module = vto.OcRFViewTransformerFull(
            pc_range=list(cfg.pc_range), bev_h=Y, bev_w=X, num_height=cfg.num_height, grid_config=cfg.grid,
            input_size=cfg.input_size, downsample=cfg.downsample, in_channels=256, out_channels=cfg.channels,
            accelerate=accelerate).to(self.device).eval()
        # strands on side streams pay off inside a captured graph (parallel branches, no host cost); issued
        # eagerly the extra stream / event calls cost more than the overlap returns (1.58 -> 1.81 ms)
        self._graph_parallel = parallel_branches
        r = synthetic.rig(cfg.n_cams, cfg.input_size, self.batch)
        Hf, Wf = cfg.feat_hw
        g = torch.Generator().manual_seed(seed)
        x = torch.randn(self.batch, cfg.n_cams, 256, Hf, Wf, generator=g)
        raw = torch.randint(0, 256, (self.batch, cfg.n_cams, 3, *cfg.input_size), generator=g).float()
        inp = [x] + [torch.from_numpy(r[k]) for k in ('rots', 'trans', 'intrins', 'post_rots', 'post_trans', 'bda')]
        inp += [torch.zeros(self.batch, cfg.n_cams, 27), raw, raw, raw, torch.from_numpy(r['c2w'])]
        # host_calibration: the six calibration tensors and c2w stay host tensors (what the dataloader produced);
        # the module then needs no device -> host read-back per forward (OcRFViewTransformerFull._to_host)
        keep = {1, 2, 3, 4, 5, 6, 11} if host_calibration else set()
        self.inputs = [t if i in keep else t.to(self.device) for i, t in enumerate(inp)]
        pre = torch.randn(self.batch * cfg.n_cams, cfg.D + 2 + cfg.channels, Hf, Wf, generator=g)
        pre[:, :cfg.D] *= 3                                       # stand-in for the DepthNet output
        self.depthnet_out = pre.to(self.device)

    @torch.no_grad()
    def step(self):
        m = self.module
        depth, fdepth, sem, feat_cl = self._ops.prefilter(self.depthnet_out, m.D, m.out_channels, m.depth_threshold,
                                                          m.semantic_threshold)
        return m.view_transform(self.inputs, fdepth, None, feat_cl)

    # ---- the same step as ONE hipGraph launch (cached geometry only) -------------------------------
    def capture(self, warmup=3):
        """Capture ``step`` into a hipGraph (view_transformer_ocrf.GraphedNeck).  The only per-step host
        decision — the random camera of each sample (view_transformer_ocrf.py:1081) — lives in two small
        static device tensors that ``step_graphed`` refreshes before every replay."""
        from .view_transformer_ocrf import GraphedNeck
        self._graphed = GraphedNeck(self.module, self.inputs, self.depthnet_out, warmup=warmup,
                                    parallel_branches=self._graph_parallel,
                                    capture_stream=getattr(self, '_capture_stream', None))
        return self

    def step_graphed(self, cam_idx_list=None):
        """-> (bev_feat, depth, bev_mask_logit, extras) like ``step`` (static tensors, overwritten by the next
        replay)."""
        bev, depth, (bev_mask, _sem), extras = self._graphed.replay(cam_idx_list)
        return bev, depth, bev_mask, extras

    @property
    def bev_voxels_per_step(self):
        X, Y, Z = self.cfg.bev_xyz
        return self.batch * Z * Y * X

    @property
    def views_per_step(self):
        return self.batch                                         # one random camera per sample (:1081)
