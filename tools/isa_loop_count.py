"""Instruction mix of the hot loop of a kernel, from hipcc's device assembly (no GPU needed):

    python tools/isa_loop_count.py rasterize.hip raster_blend_kernelILb0ELb0ELb1ELb0E v_exp_f32 8

compiles ocrfdet_amd/csrc/<file> with the library's flags (-S, device only), finds the kernel whose mangled name
contains the 2nd argument and prints VALU / SALU / LDS / VMEM counts of every basic block holding at least <n> of the
marker instruction, plus the kernel's register counts."""
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src, kern, marker, n_min = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    out = '/tmp/_isa_%s.s' % os.path.basename(src)
    subprocess.check_call(['/opt/rocm/bin/hipcc', '-O3', '--offload-arch=gfx950', '-std=c++17', '-ffp-contract=off',
                           '-I' + ROOT + '/include', '-I' + ROOT + '/ocrfdet_amd/csrc', '--offload-device-only', '-S',
                           '-o', out, os.path.join(ROOT, 'ocrfdet_amd/csrc', src)], stderr=subprocess.DEVNULL)
    lines = open(out).read().split('\n')
    start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\S*%s\S*:' % re.escape(kern), l))
    end = next(i for i in range(start, len(lines)) if lines[i].strip() == 's_endpgm')
    print(lines[start])
    for l in lines[end:end + 400]:
        if re.search(r'\.(sgpr_count|vgpr_count|vgpr_spill_count|group_segment_fixed_size)|; (NumVgprs|NumSgprs|ScratchSize|Occupancy)', l):
            print('  ', l.strip())
    blocks, cur, name = [], [], 'entry'
    for l in lines[start + 1:end]:
        t = l.strip()
        if re.match(r'^\.LBB\d+_\d+:', t):
            blocks.append((name, cur))
            name, cur = t, []
        elif t and not t.startswith((';', '.')):
            cur.append(t)
    blocks.append((name, cur))
    for name, b in blocks:
        if sum(1 for x in b if x.startswith(marker)) < n_min:
            continue
        is_salu = lambda x: x.startswith('s_') and not x.startswith(('s_waitcnt', 's_nop', 's_cbranch', 's_branch'))  # noqa: E731
        print(name, 'VALU', sum(x.startswith('v_') for x in b), 'SALU', sum(map(is_salu, b)),
              'LDS', sum(x.startswith('ds_') for x in b), 'VMEM', sum(x.startswith(('buffer_', 'global_', 'flat_')) for x in b),
              'waitcnt', sum(x.startswith('s_waitcnt') for x in b), 'nop', sum(x.startswith('s_nop') for x in b), 'all', len(b))
        c = collections.Counter(x.split()[0] for x in b)
        print('   ', ', '.join('%s %d' % kv for kv in sorted(c.items(), key=lambda kv: -kv[1])))


if __name__ == '__main__':
    main()
