"""CPU-only: the L3 wrappers (bayer / debayer / denoise / local_contrast / tonemap / color_conversion / white_balance / jpeg --
the API BASELINE.json's north_star says stays unchanged) keep the reference's signatures: every function, class, method,
argument name, order, keyword-only split and default of tests/golden/wrapper_surface.json (parsed from the reference's
sources by tests/golden/make_wrapper_surface.py) is found in this repo's wrappers, parsed the same way.  What this repo adds
is a closed list below: defaulted keyword arguments and helper methods of the fused hand-overs."""

import importlib.util
import json
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
GOLD = json.loads((ROOT / 'tests' / 'golden' / 'wrapper_surface.json').read_text())
spec = importlib.util.spec_from_file_location('make_wrapper_surface', ROOT / 'tests' / 'golden' / 'make_wrapper_surface.py')
mws = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mws)
MINE = mws.surface(ROOT / 'torch-darktable_amd' / 'torch_darktable')

# (module, function or Class.method) -> keyword arguments this repo adds (all defaulted; the fused stage hand-overs of DESIGN.md 1)
ALLOWED_EXTRA_ARGS = {
    ('denoise', 'Wiener.process_log_luminance'): {'luminance_out'},
    ('local_contrast', 'Bilateral.process_rgb'): {'luminance', 'metrics'},
    ('local_contrast', 'Bilateral.process_log_rgb'): {'luminance', 'metrics'},
    ('white_balance', 'estimate_white_balance'): {'literal_positions'},
}
# names this repo adds to a module or class
ALLOWED_EXTRA_NAMES = {
    'debayer': {'RCD.process_packed', 'RCD.process_packed12'},
    'denoise': {'Wiener.process_log_luminance_lab'},
    'local_contrast': {'Bilateral.process_lab'},
    'tonemap': {'MetricsAccumulator', 'compute_image_metrics_into'},
}


def _norm_default(d):
    """Defaults as source text, normalised for spelling only (quotes, float literals)."""
    if d is None:
        return None
    try:
        return repr(eval(d, {'__builtins__': {}}, {}))  # literals only; names fall through
    except Exception:  # noqa: BLE001
        return d.replace('"', "'")


def _check_signature(where, ref, mine, extra):
    n = len(ref['args'])
    assert mine['args'][:n] == ref['args'], f'{where}: positional arguments {mine["args"]} != reference {ref["args"]}'
    assert [_norm_default(d) for d in mine['defaults'][:n]] == [_norm_default(d) for d in ref['defaults']], (
        f'{where}: defaults {mine["defaults"][:n]} != reference {ref["defaults"]}')
    added = set(mine['args'][n:]) | (set(mine['kwonly']) - set(ref['kwonly']))
    assert added <= extra, f'{where}: undeclared extra arguments {sorted(added - extra)}'
    for a in mine['args'][n:]:  # an added positional-or-keyword argument must be defaulted
        assert mine['defaults'][mine['args'].index(a)] is not None, f'{where}: added argument {a} has no default'
    for k, d in zip(ref['kwonly'], ref['kwdefaults']):
        assert k in mine['kwonly'], f'{where}: keyword-only argument {k} missing'
        assert _norm_default(mine['kwdefaults'][mine['kwonly'].index(k)]) == _norm_default(d), f'{where}: default of {k}'
    for k in set(mine['kwonly']) - set(ref['kwonly']):
        assert mine['kwdefaults'][mine['kwonly'].index(k)] is not None, f'{where}: added keyword {k} has no default'
    assert mine['vararg'] == ref['vararg'] and mine['kwarg'] == ref['kwarg'], f'{where}: *args / **kwargs differ'
    assert mine['decorators'] == ref['decorators'], f'{where}: {mine["decorators"]} != reference {ref["decorators"]}'


@pytest.mark.parametrize('module', sorted(GOLD['modules']))
def test_wrapper_module_matches_the_reference(module):
    ref, mine = GOLD['modules'][module], MINE['modules'][module]
    for name, sig in ref['functions'].items():
        assert name in mine['functions'], f'{module}.{name} missing'
        _check_signature(f'{module}.{name}', sig, mine['functions'][name], ALLOWED_EXTRA_ARGS.get((module, name), set()))
    for cname, cls in ref['classes'].items():
        assert cname in mine['classes'], f'{module}.{cname} missing'
        mcls = mine['classes'][cname]
        assert [b.split('.')[-1] for b in mcls['bases']] == [b.split('.')[-1] for b in cls['bases']], f'{module}.{cname}: bases {mcls["bases"]} != {cls["bases"]}'
        for mname, sig in cls['methods'].items():
            assert mname in mcls['methods'], f'{module}.{cname}.{mname} missing'
            _check_signature(f'{module}.{cname}.{mname}', sig, mcls['methods'][mname], ALLOWED_EXTRA_ARGS.get((module, f'{cname}.{mname}'), set()))
        for fname, default in cls['fields'].items():
            assert fname in mcls['fields'], f'{module}.{cname}.{fname} (field) missing'
            # enum members that wrap the extension's enum are spelled through this repo's module; plain values must agree
            if default is not None and 'extension' not in default:
                assert _norm_default(mcls['fields'][fname]) == _norm_default(default), f'{module}.{cname}.{fname}: {mcls["fields"][fname]} != {default}'
    # nothing public beyond the closed list
    allowed = ALLOWED_EXTRA_NAMES.get(module, set())
    extra = set(mine['functions']) - set(ref['functions'])
    extra |= set(mine['classes']) - set(ref['classes'])
    for cname in set(mine['classes']) & set(ref['classes']):
        extra |= {f'{cname}.{m}' for m in set(mine['classes'][cname]['methods']) - set(ref['classes'][cname]['methods'])}
    assert extra <= allowed, f'{module}: undeclared public names {sorted(extra - allowed)}'


def test_package_exports_every_reference_name():
    missing = set(GOLD['exports']) - set(MINE['exports'])
    assert not missing, f'torch_darktable/__init__.py does not export {sorted(missing)}'
