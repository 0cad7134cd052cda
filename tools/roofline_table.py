"""Per-kernel table of the step from the files under profiles/ (no GPU needed): rocprofv3 kernel stats of the
--no-overlap run (every kernel alone on the device) joined with the PMC means of the same command.
    python tools/roofline_table.py [tag] > profiles/<tag>_kernel_table.md

Columns: launches per step, mean duration alone, HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE KiB: the gfx950
correction of MI355X_MICROARCH.md), achieved HBM GB/s and its share of 8 TB/s, VALU busy (SQ_ACTIVE_INST_VALU x 4 /
(1024 SIMDs x duration x 2.4 GHz)), MFMA busy (SQ_VALU_MFMA_BUSY_CYCLES / the same denominator), share of the wave
cycles spent waiting (SQ_WAIT_ANY / SQ_WAVE_CYCLES), LDS bank conflicts per LDS instruction."""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r3'
P = os.path.join(ROOT, 'profiles')
CLOCK, SIMDS, HBM = 2.4e9, 1024, 8000.0

stats = list(csv.DictReader(open(os.path.join(P, f'{tag}_cfg2_kernel_stats_serial.csv'))))
pmc = {}
for r in csv.DictReader(open(os.path.join(P, f'{tag}_pmc_mean.csv'))):
    pmc.setdefault(r['kernel'], {})[r['counter']] = (float(r['mean']), int(r['launches']))
meta = json.load(open(os.path.join(P, f'{tag}_pmc_meta.json')))
steps = max(int(r['calls']) for r in stats if 'raster_blend_sorted' in r['kernel'])
# (+ the instrumented blend launch after the timed region)
steps -= 1 if any('raster_blend_sorted' in r['kernel'] and ', true>' in r['kernel'] for r in stats) else 0

print(f'# Kernels of one cfg2 step, each alone on the device (`profiles/{tag}_cfg2_kernel_stats_serial.csv` + '
      f'`{tag}_pmc_mean.csv`; sources {meta["source_hash"]}, {steps} steps traced)\n')
print('| kernel | launches / step | mean µs | HBM MB / launch | GB/s | of 8 TB/s | VALU busy | MFMA busy | waiting | LDS conflicts / LDS instr |')
print('|---|---|---|---|---|---|---|---|---|---|')
total = 0.0
for r in stats:
    calls = int(r['calls'])
    # plan builds, uploads, warm-up variants are not part of a step; a kernel of the step may have a launch or two more
    # than steps x k (bench.py's instrumented blend launch after the timed region re-runs the update once)
    if calls < steps or calls - steps * round(calls / steps) not in (0, 1, 2):
        continue
    name = r['kernel']
    if name.startswith(('rocprim::', 'at::native', 'void at::native', '__amd_rocclr')):      # torch's own (plan building of the bench's other legs)
        continue
    us = float(r['mean_ns']) / 1e3
    per_step = round(calls / steps)
    total += us * per_step
    c = {k: v[0] for k, v in pmc.get(name, {}).items()}
    cyc = SIMDS * us * 1e-6 * CLOCK

    def f(x, fmt):
        return fmt % x if x is not None else '—'
    mb = (2 * c['FETCH_SIZE'] + c['WRITE_SIZE']) * 1024 / 1e6 if 'FETCH_SIZE' in c and 'WRITE_SIZE' in c else None
    gbs = mb * 1e6 / (us * 1e-6) / 1e9 if mb is not None else None
    valu = c['SQ_ACTIVE_INST_VALU'] * 4 / cyc if 'SQ_ACTIVE_INST_VALU' in c else None
    mfma = c['SQ_VALU_MFMA_BUSY_CYCLES'] / cyc if c.get('SQ_VALU_MFMA_BUSY_CYCLES') else None
    wait = c['SQ_WAIT_ANY'] / c['SQ_WAVE_CYCLES'] if c.get('SQ_WAVE_CYCLES') else None
    lds = c['SQ_LDS_BANK_CONFLICT'] / c['SQ_INSTS_LDS'] if c.get('SQ_INSTS_LDS') else None
    short = name.replace('(anonymous namespace)::', '').replace('void ', '')
    print(f'| `{short[:58]}` | {per_step:.1f} | {us:.1f} | {f(mb, "%.1f")} | {f(gbs, "%.0f")} | '
          f'{f(gbs / HBM if gbs is not None else None, "%.2f")} | {f(valu, "%.2f")} | {f(mfma, "%.2f")} | {f(wait, "%.2f")} | {f(lds, "%.2f")} |')
print(f'\nSum of the solo durations per step: {total:.0f} µs; the overlapped step (two HIP streams) takes '
      f'what `profiles/{tag}_cfg2_bench.json` reports.')
