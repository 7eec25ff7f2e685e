#!/usr/bin/env python
"""Instruction mix of one kernel of a `hipcc -S --cuda-device-only` listing: per basic block (label), counts by class -- the CPU-side development loop
for the VALU-bound row kernels (hyper-connections): how many instructions does the token loop issue, and of which kind.
usage: scripts/isa_count.py <file.s> <substring of the mangled kernel name> [--blocks]"""
import collections
import re
import sys


def classify(m):
    if m.startswith('v_mfma') or m.startswith('v_smfma'):
        return 'mfma'
    if m.startswith('v_pk_'):
        return 'valu_pk'
    if m in ('v_readlane_b32', 'v_readfirstlane_b32', 'v_writelane_b32'):
        return 'valu_lane'
    if m.startswith('v_accvgpr'):
        return 'valu_acc'
    if m.startswith(('v_exp', 'v_rcp', 'v_rsq', 'v_log', 'v_sqrt', 'v_sin', 'v_cos')):
        return 'valu_trans'
    if m.startswith('v_'):
        return 'valu'
    if m.startswith('s_waitcnt') or m.startswith('s_nop') or m.startswith('s_barrier'):
        return 'sync'
    if m.startswith('s_load') or m.startswith('s_buffer_load'):
        return 'smem'
    if m.startswith('s_'):
        return 'salu'
    if m.startswith('ds_'):
        return 'lds'
    if m.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
        return 'vmem'
    return 'other'


def main():
    path, key = sys.argv[1], sys.argv[2]
    show_blocks = '--blocks' in sys.argv
    lines = open(path).read().split('\n')
    start = None
    for i, l in enumerate(lines):
        if l.startswith('_Z') and ':' in l and key in l.split(':')[0]:
            start = i
            break
    if start is None:
        raise SystemExit(f'no kernel label containing {key!r}')
    blocks, cur = collections.OrderedDict(), 'entry'
    blocks[cur] = collections.Counter()
    meta = {}
    for l in lines[start + 1:]:
        if l.startswith('.Lfunc_end'):
            break
        s = l.strip()
        if not s or s.startswith(';'):
            m = re.match(r';\s*(NumVgprs|NumAgprs|NumSgprs|ScratchSize|Occupancy|sgpr_spill_count|vgpr_spill_count|codeLenInByte):\s*(\d+)', s)
            if m:
                meta[m.group(1)] = int(m.group(2))
            continue
        if re.match(r'^\.LBB[\w]+:', s):
            cur = s.split(':')[0]
            blocks[cur] = collections.Counter()
            continue
        if s.startswith('.'):
            continue
        mn = s.split()[0]
        blocks[cur][classify(mn)] += 1
        blocks[cur]['_' + mn] += 1
    # meta sits after the function end
    for l in lines[start:start + 200000]:
        m = re.match(r';\s*(NumVgprs|NumAgprs|NumSgprs|ScratchSize|Occupancy|sgpr_spill_count|vgpr_spill_count|codeLenInByte):\s*(\d+)', l.strip())
        if m and m.group(1) not in meta:
            meta[m.group(1)] = int(m.group(2))
        if l.startswith('.Lfunc_end') and len(meta) >= 6:
            pass
        if len(meta) >= 8:
            break
    tot = collections.Counter()
    for b in blocks.values():
        tot.update({k: v for k, v in b.items() if not k.startswith('_')})
    print(lines[start][:120])
    print('meta', meta)
    print('whole kernel', dict(tot), 'sum', sum(tot.values()))
    big = sorted(blocks.items(), key=lambda kv: -sum(v for k, v in kv[1].items() if not k.startswith('_')))[:6]
    for name, c in big:
        n = sum(v for k, v in c.items() if not k.startswith('_'))
        print(f'block {name}: {n} instr', {k: v for k, v in c.items() if not k.startswith('_')})
        if show_blocks:
            top = sorted(((k[1:], v) for k, v in c.items() if k.startswith('_')), key=lambda kv: -kv[1])[:28]
            print('    ', top)


if __name__ == '__main__':
    main()
