"""Host logic of the panel pooling plan (ocrfdet_amd/bevpool.MfmaPoolPlan) on the CPU: the plan is index algebra on torch
tensors, so it builds on CPU tensors once the device check is patched out (no kernel runs here).  Checked: every panel
belongs to exactly one unit, a tile's units are its slices 0..n-1, units respect ``group`` and ``unit_cost``, the cells of a
panel are voxel-major with ``panel_voff`` as their index, and walking the plan the way csrc/bev_pool_panel.hip does
(cell weights, then per voxel the cells in order) reproduces the pooling of bev_pool_cuda.cu:39-47."""
import numpy as np
import pytest
import torch

from ocrfdet_amd import _lib, bevpool


@pytest.fixture()
def cpu_plans(monkeypatch):
    monkeypatch.setattr(_lib, 'require_cuda', lambda *a: None)
    return bevpool.MfmaPoolPlan


def _problem(seed, n, n_vox, n_rows, n_depth=1000):
    rng = np.random.default_rng(seed)
    w = rng.pareto(1.2, n_vox) + 0.05                                  # skewed voxel populations
    rb = np.sort(rng.choice(n_vox, size=n, p=w / w.sum())).astype(np.int32)
    rf = rng.integers(0, n_rows, n).astype(np.int32)
    rd = rng.integers(0, n_depth, n).astype(np.int32)
    depth = rng.random(n_depth).astype(np.float32)
    feat = rng.standard_normal((n_rows, 80)).astype(np.float32)
    return rb, rf, rd, depth, feat


@pytest.mark.parametrize('group,unit_cost', [(8, None), (2, None), (8, 2.5), (3, 1.0)])
def test_plan_partition_order_and_emulated_pooling(cpu_plans, group, unit_cost):
    B, Z, Y, X, C = 1, 2, 13, 21, 80                                   # ragged grid: tiles cut by the border
    rb, rf, rd, depth, feat = _problem(group, 9000, B * Z * Y * X, 300)
    plan = cpu_plans(torch.from_numpy(rd), torch.from_numpy(rf), torch.from_numpy(rb), (B, Z, Y, X, C), group=group,
                     unit_cost=unit_cost)
    KP = plan.panel_rows.numel() // max(plan.n_panels, 1)
    units = plan.units.numpy()
    coff = plan.panel_cell_off.numpy()
    # partition: every panel once, slices of a tile consecutive, group respected
    seen = np.zeros(plan.n_panels, int)
    for t, p0, p1, sl in units:
        seen[p0:p1] += 1
        assert p1 - p0 <= group
    assert (seen == 1).all()
    assert (np.bincount(units[:, 0], minlength=plan.n_tiles) >= 1).all()            # empty tiles have a unit too
    for t in np.unique(units[:, 0]):
        us = units[units[:, 0] == t]
        ns = us[0, 3] >> 16
        assert len(us) == ns and sorted((us[:, 3] & 0xFFFF).tolist()) == list(range(ns))
        order = np.argsort(us[:, 3] & 0xFFFF)
        assert (us[order][1:, 1] == us[order][:-1, 2]).all()                       # slice s + 1 starts where slice s ends
    per_tile = {int(u[0]): int(u[3] >> 16) for u in units}
    assert plan.n_slab_slices == sum(n for n in per_tile.values() if n > 1)        # a slab per unit of every cut tile
    # cells: voxel-major inside a panel, panel_voff indexes them
    code = plan.cell_code.numpy().astype(np.int64) & 0xFFFF
    voff = plan.panel_voff.numpy().reshape(-1, 64)
    for p in range(plan.n_panels):
        c = code[coff[p]:coff[p + 1]]
        key = (c >> 8) * 256 + (c & 0xFF)
        assert (np.diff(key) > 0).all()                                              # (voxel, row) strictly ascending
        starts = np.searchsorted(c >> 8, np.arange(64))
        assert (voff[p] == starts).all()
        assert ((c & 0xFF) < plan.panel_nrows.numpy()[p]).all()
    # the kernel's walk on the CPU: weights per cell, then per voxel its cells in order
    cells, rds = plan.cells.numpy(), plan.rd_sorted.numpy()
    cw = np.zeros(plan.n_cells, np.float64)
    for c in range(plan.n_cells):
        x, y, z, w = cells[c]
        npt = (x >> 16) & 0xFFFF
        idx = [y, z, w][:npt] if npt <= 3 else list(rds[y:y + z])
        cw[c] = depth[idx].astype(np.float64).sum()
    rows = plan.panel_rows.numpy().reshape(-1, KP)
    tx = (X + 7) // 8
    tpp = tx * ((Y + 7) // 8)
    out = np.zeros((B * Z * Y * X, C))
    for t, p0, p1, sl in units:
        plane, kt = t // tpp, t % tpp
        y0, x0 = (kt // tx) * 8, (kt % tx) * 8
        for p in range(p0, p1):
            for c in range(coff[p], coff[p + 1]):
                v, r = code[c] >> 8, code[c] & 0xFF
                out[plane * Y * X + (y0 + v // 8) * X + x0 + v % 8] += cw[c] * feat[rows[p][r]]
    want = np.zeros_like(out)
    np.add.at(want, rb, depth[rd][:, None].astype(np.float64) * feat[rf])
    np.testing.assert_allclose(out, want, rtol=1e-9, atol=1e-9)


def test_unit_cost_cuts_long_runs_finer(cpu_plans):
    """A tile whose panels hold long voxel runs (one voxel seen by many rows) gets more, shorter units under ``unit_cost``."""
    B, Z, Y, X, C = 1, 1, 8, 8, 80
    n_rows = 480                                                       # ten panels of 48 rows
    rb = np.zeros(n_rows, np.int32)                                    # every row hits voxel 0: runs of 48 cells per panel
    rf = np.arange(n_rows, dtype=np.int32)
    rd = np.zeros(n_rows, np.int32)
    mk = lambda **kw: cpu_plans(torch.from_numpy(rd), torch.from_numpy(rf), torch.from_numpy(rb), (B, Z, Y, X, C), **kw)  # noqa: E731
    a, b = mk(group=8), mk(group=8, unit_cost=8.0)
    assert a.n_panels == b.n_panels == 10
    assert a.n_units == 2 and b.n_units > a.n_units                    # cost of a panel = 1 + 48 / 8 = 7: one or two per unit
    assert (b.units.numpy()[:, 2] - b.units.numpy()[:, 1]).max() <= 2


def test_empty_ranks_give_one_empty_unit_per_tile(cpu_plans):
    e = torch.zeros(0, dtype=torch.int32)
    plan = cpu_plans(e, e, e, (1, 1, 16, 16, 80))
    assert plan.n_units == plan.n_tiles == 4 and plan.n_points == 0
    assert (plan.units.numpy()[:, 1] == plan.units.numpy()[:, 2]).all()
