"""Static instruction mix of every gfx950 kernel in libtdk_hip.so (from the code objects' disassembly).

  python profiles/isa_mix.py [--json] [--filter SUBSTR]

Per kernel: VALU instructions, of which transcendental (v_exp/v_log/v_rcp/v_rsq/v_sqrt/v_sin/v_cos: 4 issue
cycles per wave64 instead of 2 on a SIMD-32 -- MI355X_MICROARCH.md 'vector-instruction ISSUE cost'), DPP /
cross-lane VALU, SALU, LDS, VMEM, plus VGPRs / LDS bytes from the kernel descriptor notes.  Static counts:
loops and branches are not weighted."""
import json
import re
import shutil
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
LIB = ROOT / 'torch-darktable_amd' / 'torch_darktable' / 'libtdk_hip.so'
LLVM = Path('/opt/rocm/lib/llvm/bin')
TRANS = re.compile(r'^v_(exp|log|rcp|rsq|sqrt|sin|cos)_')
# half-rate classes measured on gfx950 (tests/hip_unit/valu_issue_bench.hip): any SGPR / VCC source operand, the 3-input
# min/max/med forms, conversions to f16
SGPR_OP = re.compile(r'(?<![\w.])(s\d+|s\[\d+:\d+\]|vcc|exec|m0)(?![\w])')
HALF3 = re.compile(r'^v_(max3|min3|med3)_')


def demangle(names):
    for tool in (str(LLVM / 'llvm-cxxfilt'), 'c++filt'):
        try:
            out = subprocess.run([tool], input='\n'.join(names), capture_output=True, text=True).stdout.splitlines()
            if len(out) == len(names):
                return dict(zip(names, out))
        except Exception:  # noqa: BLE001
            pass
    return {n: n for n in names}


def main():
    as_json = '--json' in sys.argv
    flt = sys.argv[sys.argv.index('--filter') + 1] if '--filter' in sys.argv else ''
    mix = {}
    with tempfile.TemporaryDirectory() as td:
        lib = Path(td) / 'lib.so'
        shutil.copy(LIB, lib)
        subprocess.run([str(LLVM / 'llvm-objdump'), '--offloading', str(lib)], capture_output=True, cwd=td)
        for co in sorted(Path(td).glob('lib.so.*gfx950*')):
            dis = subprocess.run([str(LLVM / 'llvm-objdump'), '-d', '--mcpu=gfx950', str(co)], capture_output=True, text=True).stdout
            cur = None
            for line in dis.splitlines():
                m = re.match(r'^[0-9a-f]+ <(.+)>:$', line)
                if m:
                    cur = m.group(1)
                    mix[cur] = {'valu': 0, 'trans': 0, 'packed': 0, 'dpp': 0, 'salu': 0, 'lds': 0, 'vmem': 0, 'mfma': 0, 'half': 0, 'cnd_vcc': 0, 'div_fmas': 0}
                    continue
                if cur is None:
                    continue
                parts = line.strip().split()
                if not parts:
                    continue
                op = parts[0]
                d = mix[cur]
                if op.startswith('v_mfma'):
                    d['mfma'] += 1
                elif op.startswith('v_'):
                    d['valu'] += 1
                    if TRANS.match(op):
                        d['trans'] += 1
                    elif op.startswith(('v_pk_fma_f32', 'v_pk_add_f32', 'v_pk_mul_f32')):
                        d['packed'] += 1  # two fp32 operations per lane: 4 SIMD cycles (tests/hip_unit/pk_issue_bench.hip)
                    elif op.startswith('v_cndmask_b32_e32') or op.startswith('v_cndmask_b32_dpp'):
                        d['cnd_vcc'] += 1  # VOP2 select reading VCC: ~23 cycles
                    elif op.startswith('v_div_fmas'):
                        d['div_fmas'] += 1
                    else:
                        srcs = line.strip().split(None, 1)[1].split(',', 1)[1] if ',' in line else ''
                        if HALF3.match(op) or op.startswith('v_cvt_f16_f32') or SGPR_OP.search(srcs.split('//')[0]):
                            d['half'] += 1
                    if 'dpp' in line or op.startswith(('v_permlane', 'v_readlane', 'v_writelane', 'v_readfirstlane')):
                        d['dpp'] += 1
                elif op.startswith('ds_'):
                    d['lds'] += 1
                elif op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
                    d['vmem'] += 1
                elif op.startswith('s_'):
                    d['salu'] += 1
    names = demangle(list(mix))
    mix = {names[k]: v for k, v in mix.items() if v['valu'] + v['salu'] > 0 and flt in names[k]}
    if as_json:
        print(json.dumps(mix))
        return
    for k, v in sorted(mix.items(), key=lambda kv: -kv[1]['valu']):
        full = v['valu'] - v['trans'] - v['packed'] - v['half'] - v['cnd_vcc'] - v['div_fmas']
        est = full * 2.4 + (v['half'] + v['packed']) * 4.8 + v['trans'] * 8.5 + (v['cnd_vcc'] + v['div_fmas']) * 23
        print(f"{v['valu']:6d} valu ({v['trans']:4d} trans, {v['packed']:4d} packed, {v['half']:4d} half-rate, {v['cnd_vcc'] + v['div_fmas']:3d} vcc-read, {v['dpp']:3d} xlane) ~{est:8.0f} issue cyc "
              f"{v['salu']:5d} salu {v['lds']:5d} lds {v['vmem']:4d} vmem  {k[:110]}")


if __name__ == '__main__':
    main()
