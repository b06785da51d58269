"""CPU-only (hipcc cross-compiles): instruction-level assumptions the kernels' comments rely on, checked in the gfx950 ISA.

1. tdk_image_metrics (csrc/tonemap.hip, metrics_kernel<T, true>): the one-launch form hands the per-workgroup sums to the
   workgroup that draws the last ticket without release / acquire fences (a fence per workgroup writes the XCD's L2 back: 20 us
   instead of 10).  That is only sound while (a) the row adds are hardware float atomics -- global_atomic_add_f32, executed at
   the memory side, nothing of them left in the issuing XCD's L2 -- and not a compare-and-swap loop on cached data, (b) the
   ticket is a returning integer atomic, and (c) the last workgroup reads the rows with agent-scope loads (sc1: they bypass the
   non-coherent levels).  If a compiler change lowers any of these differently this test fails instead of the metrics going
   silently stale; tests/test_gpu_fusion.py::test_metrics_one_launch_on_many_grids is the run-time side of the same contract.
2. RCD strips, approximate flavour (csrc/tdk_rcd_stream.h): no IEEE division expansion (v_div_scale / v_div_fmas / v_div_fixup)
   is left in the float16 kernels' inner blocks -- the instruction saving the flavour exists for."""
import re
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
CSRC = ROOT / 'torch-darktable_amd' / 'csrc'
HIPCC = '/opt/rocm/bin/hipcc'
FLAGS = ['-O3', '-std=c++17', '--offload-arch=gfx950', '-ffp-contract=off', '-fno-slp-vectorize', '--cuda-device-only', '-S', '-o', '-']


def _asm(stem):
    r = subprocess.run([HIPCC, *FLAGS, str(CSRC / f'{stem}.hip')], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout


def _kernels(asm, pattern):
    """{symbol: body lines} of the kernels whose mangled name matches."""
    out, cur, name = {}, None, None
    for line in asm.split('\n'):
        m = re.match(r'^(_Z\w+):', line)
        if m and re.search(pattern, m.group(1)):
            name, cur = m.group(1), []
            continue
        if cur is not None:
            cur.append(line.strip())
            if line.strip().startswith('s_endpgm'):
                out[name] = cur
                cur = None
    return out


@pytest.fixture(scope='module')
def tonemap_asm():
    return _asm('tonemap')


def test_metrics_one_launch_handoff_instructions(tonemap_asm):
    ks = _kernels(tonemap_asm, r'metrics_kernelI\w+Lb1E')
    assert len(ks) == 2, list(ks)  # float and __half storage
    for name, body in ks.items():
        ops = [l.split()[0] for l in body if l and not l.startswith((';', '.'))]
        assert 'global_atomic_add_f32' in ops, f'{name}: the row adds are no longer global_atomic_add_f32'
        assert not any(o.startswith('global_atomic_cmpswap') for o in ops), f'{name}: a compare-and-swap loop appeared'
        ticket = [i for i, l in enumerate(body) if l.startswith('global_atomic_add ') and 'sc0' in l]
        assert len(ticket) == 1, f'{name}: expected one returning integer ticket atomic, found {len(ticket)}'
        tail_loads = [l for l in body[ticket[0]:] if l.startswith('global_load_dword')]
        assert len(tail_loads) >= 6 and all(' sc1' in l for l in tail_loads), f'{name}: row loads of the last workgroup without agent scope: {tail_loads[:3]}'
        # the adds complete before the ticket is drawn
        between = body[body.index(next(l for l in body if l.startswith('global_atomic_add_f32'))):ticket[0]]
        assert any(l.startswith('s_waitcnt vmcnt(0)') for l in between) and any(l.startswith('s_barrier') for l in between), name


def test_bounds_one_launch_handoff_instructions(tonemap_asm):
    """tdk_image_bounds (bounds_ticket_kernel): the same fence-less hand-off as the metrics kernel.  Minimum and maximum are single
    integer atomics on order-preserving keys (no compare-and-swap loop on cached data), they have completed (vmcnt(0)) before the
    returning ticket atomic is issued, and the workgroup that draws the last ticket reads the two keys with agent-scope loads."""
    ks = _kernels(tonemap_asm, r'bounds_ticket_kernelI')
    assert len(ks) == 2, list(ks)
    for name, body in ks.items():
        ops = [l.split()[0] for l in body if l and not l.startswith((';', '.'))]
        assert ops.count('global_atomic_umin') == 1 and ops.count('global_atomic_umax') == 1, name
        assert not any(o.startswith('global_atomic_cmpswap') for o in ops), f'{name}: a compare-and-swap loop appeared'
        ticket = [i for i, l in enumerate(body) if l.startswith('global_atomic_add ') and 'sc0' in l]
        assert len(ticket) == 1, name
        umax = next(i for i, l in enumerate(body) if l.startswith('global_atomic_umax'))
        assert umax < ticket[0] and any(l.startswith('s_waitcnt vmcnt(0)') for l in body[umax:ticket[0]]), name
        tail_loads = [l for l in body[ticket[0]:] if l.startswith('global_load_dword')]
        assert len(tail_loads) == 2 and all(' sc1' in l for l in tail_loads), f'{name}: key loads without agent scope: {tail_loads}'


def test_rcd_approximate_flavour_has_no_ieee_division_in_its_inner_blocks():
    asm = _asm('rcd')
    ks = _kernels(asm, r'(rcd_quadILi4E\w+Lb1E|rcd_streamI\w+Lb1E)')
    assert len(ks) == 4, list(ks)  # {quad, stream} x {half, float mosaic} with float16 results
    for name, body in ks.items():
        ops = [l.split()[0] for l in body if l and not l.startswith((';', '.'))]
        n_rcp = sum(o.startswith('v_rcp_f32') for o in ops)
        assert n_rcp >= 30, (name, n_rcp)
        # the only IEEE divisions left are the ring pieces' three-sample averages (border_average) and stale_diff's index arithmetic
        n_fix = sum(o.startswith('v_div_fixup_f32') for o in ops)
        assert n_fix <= 12, f'{name}: {n_fix} IEEE divisions in the approximate flavour'
