"""The hand-counted wait of hc_bwd's LDS-DMA variant, checked against the compiler's output (CPU test: `hipcc -S` cross-compiles gfx950 here).

hc_bwd_kernel<..., GL = true> (csrc/hyper.hip) waits for the DMA group that filled the current LDS buffer with `s_waitcnt vmcnt(15)`: 15 = the previous
token's 6 stores (S x dR, dy, dbeta) + the 9 DMA instructions of the group just issued.  vmcnt retires in order, so the wait is exact only if that many
vector-memory operations REALLY sit between two groups.  More of them (a spill the compiler adds) would make the wait longer than needed but still safe;
FEWER would let the kernel read a buffer that has not landed.  This test reads the ISA of the production instantiation and asserts, per token body:
exactly 9 `global_load_lds`, then `s_waitcnt vmcnt(15)`, then >= 6 `global_store` before the next group -- and that the kernel fits three workgroups per
CU without scratch (168 VGPRs), which is the point of the variant."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
KERNEL = 'hc_bwd_kernelItLi4ELi4ELb1ELb1ELb1ELb1ELi0ELb1E'       # <bf16, S = 4, WPT = 4, WIDTH, DEPTH, LNF, PF, BC = 0, GL>


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which('hipcc')), reason='needs hipcc')
def test_hc_bwd_lds_dma_variant_keeps_the_counted_wait_contract(tmp_path):
    out = str(tmp_path / 'hyper.s')
    src = os.path.join(ROOT, 'audiolm-pytorch_amd', 'csrc', 'hyper.hip')
    subprocess.run([HIPCC if os.path.exists(HIPCC) else 'hipcc', '-S', '--cuda-device-only', '--offload-arch=gfx950', '-O3', '-std=c++17', src, '-o', out],
                   check=True, stderr=subprocess.DEVNULL)
    text = open(out).read()
    m = re.search(r'^(_Z\w*' + KERNEL + r'\w*):.*?s_endpgm', text, re.S | re.M)
    assert m, 'the GL instantiation of hc_bwd_kernel is in the library'
    body = m.group(0).split('\n')
    meta = re.search(r'\.amdhsa_kernel ' + re.escape(m.group(1)) + r'.*?\.end_amdhsa_kernel', text, re.S).group(0)
    assert int(re.search(r'\.amdhsa_next_free_vgpr (\d+)', meta).group(1)) <= 168, 'three workgroups per CU'
    assert int(re.search(r'\.amdhsa_private_segment_fixed_size (\d+)', meta).group(1)) == 0, 'no scratch (a reload in the loop drains vmcnt(0))'
    assert 3 * int(re.search(r'\.amdhsa_group_segment_fixed_size (\d+)', meta).group(1)) <= 160 * 1024, 'three workgroups of LDS per CU'
    ops = []                                                     # the vector-memory events of the kernel in program order
    for ln in body:
        t = ln.strip()
        if t.startswith('global_load_lds'):
            ops.append('dma')
        elif t.startswith(('global_store', 'buffer_store', 'scratch_store')):
            ops.append('store')
        elif t.startswith(('global_load', 'buffer_load', 'scratch_load', 'flat_load')):
            ops.append('load')
        elif t.startswith('s_waitcnt') and 'vmcnt' in t:
            ops.append('wait' + re.search(r'vmcnt\((\d+)\)', t).group(1))
    waits = [i for i, o in enumerate(ops) if o == 'wait15']
    assert len(waits) == 2, ops                                  # the token loop is unrolled twice (two buffers)
    for w in waits:
        assert ops[w - 9:w] == ['dma'] * 9, (w, ops[max(0, w - 12):w])          # the group just issued: 9 instructions, nothing else in between
    first, second = waits
    between = ops[first + 1:second - 9]                           # first token body: from its wait to the next group's first DMA
    assert between.count('store') >= 6 and 'load' not in between and not any(o.startswith('wait') for o in between), between
    after = ops[second + 1:]                                      # second token body, then (behind a full wait) the epilogue's partial-row stores
    body2 = after[:after.index('wait0')] if 'wait0' in after else after
    assert body2.count('store') >= 6 and 'load' not in body2 and 'dma' not in body2, body2
    assert 'wait0' not in ops[first:second + 8], 'no full drain inside the token loop'


RING = 'gemm_ring_kernelILi128ELi128ELi2ELi2ELb0ELb0ELi4EE'          # <128, 128, 2, 2, NT, bf16 out, NS = 4>


@pytest.mark.skipif(not (os.path.exists(HIPCC) or shutil.which('hipcc')), reason='needs hipcc')
def test_gemm_ring_kernel_keeps_its_counted_waits(tmp_path):
    """The 4-stage DMA-ring form of the 128 x 128 NT tile (csrc/gemm.hip, tile 16) orders its LDS-DMA by hand: per K-step ONE counted wait --
    `s_waitcnt vmcnt(16)` in steady state (two newer stages of 8 pieces per wave may stay in flight), 8 and 0 in the drain -- then `lgkmcnt(0)` and a raw
    `s_barrier`, then the 8 DMA pieces of stage kt + 3, then the MFMAs.  If the compiler ever put a full `vmcnt(0)` between the issue and the MFMAs (it
    does when it cannot tell the DMA destination from the buffer being read) the ring would silently degrade to prefetch distance 0; if a piece were
    added or dropped the counts 16 / 8 would be wrong and a stage could be read before it landed.  Checked on the compiler's output."""
    out = str(tmp_path / 'gemm.s')
    src = os.path.join(ROOT, 'audiolm-pytorch_amd', 'csrc', 'gemm.hip')
    subprocess.run([HIPCC if os.path.exists(HIPCC) else 'hipcc', '-S', '--cuda-device-only', '--offload-arch=gfx950', '-O3', '-std=c++17', src, '-o', out],
                   check=True, stderr=subprocess.DEVNULL)
    text = open(out).read()
    m = re.search(r'^(_Z\w*' + RING + r'\w*):.*?s_endpgm', text, re.S | re.M)
    assert m, 'the ring instantiation is in the library'
    meta = re.search(r'\.amdhsa_kernel ' + re.escape(m.group(1)) + r'.*?\.end_amdhsa_kernel', text, re.S).group(0)
    assert int(re.search(r'\.amdhsa_private_segment_fixed_size (\d+)', meta).group(1)) == 0
    ev = []
    for ln in m.group(0).split('\n'):
        t = ln.strip()
        if t.startswith('buffer_load_dwordx4') and ' lds' in t:
            ev.append('dma')
        elif t.startswith('v_mfma'):
            ev.append('mfma')
        elif t.startswith('s_barrier'):
            ev.append('bar')
        elif t.startswith('s_waitcnt') and 'vmcnt' in t:
            ev.append('wait' + re.search(r'vmcnt\((\d+)\)', t).group(1))
    assert {'wait16', 'wait8', 'wait0'} <= set(ev), sorted(set(ev))
    # program order of the (rotated) loop in the listing: prologue = NS - 1 = 3 stages x 8 pieces | the step's 16 MFMAs as ONE uninterrupted run (no
    # vector-memory wait between them: the in-flight stages stay in flight under the whole MFMA block) | the counted wait (0 / 8 / 16 by remaining stages)
    # | raw barrier | the 8 pieces of stage kt + 3 | back edge
    first_mfma = ev.index('mfma')
    assert ev[:first_mfma] == ['dma'] * 24, ev[:first_mfma]
    run = 0
    while ev[first_mfma + run] == 'mfma':
        run += 1
    assert run == 16, run
    tail = ev[first_mfma + run:]
    i16 = tail.index('wait16')
    assert all(e in ('wait0', 'wait8') for e in tail[:i16]), tail[:i16]          # the drain variants of the same wait, nothing else
    assert tail[i16 + 1] == 'bar' and tail[i16 + 2:i16 + 10] == ['dma'] * 8, tail[i16:i16 + 12]
    assert tail[i16 + 10:i16 + 12] == ['wait0', 'bar'], tail[i16 + 10:i16 + 14]  # loop exit: everything landed before the epilogue reuses the stage buffers
