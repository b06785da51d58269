"""Static view of one kernel's ISA: instruction mix per basic block (label to label), with the s_barrier positions marked.
    python profiles/isa_blocks.py file.s <kernel-name-substring> [min_instrs]
Trip counts are not known statically: weight the blocks by hand (loops are tagged with their depth comment)."""
import re
import sys

path, sub = sys.argv[1], sys.argv[2]
mn = int(sys.argv[3]) if len(sys.argv) > 3 else 1
lines = open(path).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith('_Z') and sub in l and l.rstrip().endswith(':') or (l.startswith('_Z') and sub in l and ': ' in l and '@' in l))
end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i] and all('.LBB' not in lines[j] for j in range(i + 1, min(i + 3, len(lines)))))
blocks = []
cur = {'label': 'entry', 'n': {}, 'depth': '', 'line': start}
def kind(op):
    if op.startswith('v_pk_'): return 'vpk'
    if op in ('v_exp_f32_e32', 'v_log_f32_e32', 'v_rcp_f32_e32', 'v_rsq_f32_e32', 'v_sqrt_f32_e32', 'v_exp_f32_e64', 'v_log_f32_e64', 'v_rcp_f32_e64'): return 'trans'
    if op.startswith('v_readlane') or op.startswith('v_writelane') or op.startswith('v_readfirstlane'): return 'vlane'
    if op.startswith('v_'): return 'valu'
    if op.startswith('s_load') or op.startswith('s_buffer'): return 'smem'
    if op.startswith('s_waitcnt') or op.startswith('s_nop'): return 'wait'
    if op.startswith('s_barrier'): return 'BARRIER'
    if op.startswith('s_cbranch') or op.startswith('s_branch'): return 'branch'
    if op.startswith('s_'): return 'salu'
    if op.startswith('ds_'): return 'lds'
    if op.startswith('global_') or op.startswith('buffer_') or op.startswith('flat_') or op.startswith('scratch_'): return 'vmem'
    return 'other'
for i in range(start + 1, end + 1):
    l = lines[i]
    m = re.match(r'^(\.LBB\S+):\s*(;.*)?$', l)
    if m:
        blocks.append(cur)
        d = re.search(r'Depth=(\d+)', l)
        cur = {'label': m.group(1), 'n': {}, 'depth': d.group(1) if d else '', 'line': i}
        continue
    t = l.strip()
    if not t or t.startswith(';') or t.startswith('.'):
        continue
    op = t.split()[0]
    k = kind(op)
    cur['n'][k] = cur['n'].get(k, 0) + 1
blocks.append(cur)
tot = {}
for b in blocks:
    n = sum(v for k, v in b['n'].items())
    for k, v in b['n'].items():
        tot[k] = tot.get(k, 0) + v
    if n >= mn or 'BARRIER' in b['n']:
        print(f"{b['label']:14s} L{b['line'] - start:5d} d={b['depth']:1s} n={n:4d} ", ' '.join(f'{k}={v}' for k, v in sorted(b['n'].items())))
print('total', tot)
