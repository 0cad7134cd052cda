"""Diagnostic: where the training iteration's host time comes from — forward ops grouped by the ocrfdet_amd source line that
issued them, backward nodes grouped by autograd function (torch profiler with stacks, 3 iterations).

    python tools/prof_neck_train_sources.py [config]
"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ocrfdet_amd import hotpath, synthetic  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'cfg2_6cam_2frame_bev200x200_render_hoa'
    cfg = synthetic.CONFIGS[name]
    neck = hotpath.NeckPath(cfg, torch.device('cuda:0'), accelerate=False)
    m = neck.module.train()
    pre = neck.depthnet_out
    depth0 = pre[:, :cfg.D].softmax(1)
    feat0 = pre[:, cfg.D + 2:cfg.D + 2 + cfg.channels].clone()

    def it():
        depth = depth0.clone().requires_grad_(True)
        feat = feat0.clone().requires_grad_(True)
        bev, _, logit, lst = m.view_transform_core(neck.inputs, depth, feat)
        loss = bev.square().mean() + logit.square().mean() + lst[0].mean() + lst[6].mean() + lst[4].square().mean()
        loss.backward()
        m.zero_grad(set_to_none=True)

    for _ in range(3):
        it()
    torch.cuda.synchronize()
    from torch.profiler import ProfilerActivity, profile
    n = 3
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as p:
        for _ in range(n):
            it()
        torch.cuda.synchronize()
    by_line, by_node = collections.Counter(), collections.Counter()
    t_line, t_node = collections.Counter(), collections.Counter()
    for e in p.events():
        if e.device_type != torch.autograd.DeviceType.CPU:
            continue
        if e.name.startswith('autograd::engine::evaluate_function'):
            k = e.name.split(': ')[-1]
            by_node[k] += 1
            t_node[k] += e.cpu_time_total
            continue
        if not e.name.startswith('aten::') or e.cpu_parent is not None and e.cpu_parent.name.startswith(('aten::', 'autograd::')):
            continue
        if e.cpu_parent is not None and 'Backward' in e.cpu_parent.name:
            continue
        frames = [f for f in (e.stack or []) if 'ocrfdet_amd' in f or 'tools/' in f]
        k = frames[0].split('/')[-1] if frames else '(no ocrfdet_amd frame)'
        by_line[k] += 1
        t_line[k] += e.cpu_time_total
    print(f'forward: top-level aten ops by source line (per iteration; {sum(by_line.values()) // n} ops, '
          f'{sum(t_line.values()) / n / 1e3:.2f} ms)')
    for k, c in sorted(t_line.items(), key=lambda kv: -kv[1])[:60]:
        print(f'  {c / n / 1e3:7.3f} ms  n={by_line[k] // n:4d}  {k[:140]}')
    print(f'backward: autograd nodes (per iteration; {sum(by_node.values()) // n} nodes, {sum(t_node.values()) / n / 1e3:.2f} ms)')
    for k, c in sorted(t_node.items(), key=lambda kv: -kv[1])[:50]:
        print(f'  {c / n / 1e3:7.3f} ms  n={by_node[k] // n:4d}  {k[:120]}')


if __name__ == '__main__':
    main()
