"""The drop-in boundary, pinned mechanically: every function, class, argument name, argument order, default value,
property and enum member the reference's pybind11 module registers (tests/golden/extension_surface.json, parsed from
/root/reference/torch_darktable/csrc/extension.cpp:50-248 and torch_darktable_extension.pyi by
tests/golden/make_extension_surface.py) must exist with the same spelling in torch_darktable.torch_darktable_extension.
Anything this module has BEYOND the reference's surface is listed here explicitly, must come after the reference's
arguments and must have a default, so a reference caller never sees it."""

import inspect
import json
import math
from pathlib import Path

import pytest

SURFACE = json.loads((Path(__file__).parent / 'golden' / 'extension_surface.json').read_text())

# extra trailing keyword arguments (all defaulted) of this implementation
ALLOWED_EXTRA_ARGS = {
  'estimate_white_balance': ['literal_positions'],           # default True = the reference's read pattern
}
# extra methods / functions: fused entry points and hand-overs; not part of the reference surface
ALLOWED_EXTRA_METHODS = {
  'RCD': {'process_packed12'},                         # decode12 -> white balance -> RCD as one call
  'Wiener': {'process_log_luminance', 'process_log_luminance_lab'},  # denoise.py:54-58 as one call; its Lab hand-over form
  'Bilateral': {'process_rgb', 'process_log_rgb', 'grid_size', 'process_lab'},  # local_contrast.py:109-125 as one call; grid dimensions; Lab hand-over
}
# extra module-level names: the metrics accumulator, the pipeline's normalise kernel, the reference's own create_wiener
# helper (denoise.py:112) and the exception type the reference registers
ALLOWED_EXTRA_NAMES = {'MetricsAccumulator', 'normalize_image', 'create_wiener', 'JpegException',
                       'verification_paths',  # thread-local context the GPU tests use to ask for an op's second kernel path
                       'concurrent_frames'}   # thread-local context: other frames' kernels are in flight on other streams (sharding.FrameStreams)


@pytest.fixture(scope='module')
def ext():
  import torch_darktable.torch_darktable_extension as e
  return e


def _params(fn, skip_self=False):
  ps = list(inspect.signature(fn).parameters.values())
  return ps[1:] if skip_self and ps and ps[0].name == 'self' else ps


def _check_args(label, fn, ref_args, skip_self=False, member_defaults=None):
  ps = _params(fn, skip_self)
  names = [p.name for p in ps]
  ref_names = [a[0] for a in ref_args]
  assert names[:len(ref_names)] == ref_names, f'{label}: arguments {names} do not start with the reference\'s {ref_names}'
  for p, (name, default) in zip(ps, ref_args):
    if default is None and member_defaults is not None:  # default-constructible struct: the member initialiser is the default
      assert p.default == member_defaults[name], f'{label}({name}): default {p.default!r}, reference member initialiser {member_defaults[name]!r}'
    elif default is None:
      assert p.default is inspect.Parameter.empty, f'{label}({name}): the reference has no default, here {p.default!r}'
    else:
      assert p.default is not inspect.Parameter.empty, f'{label}({name}): default {default!r} missing'
      same = (p.default == default) if not isinstance(default, float) else math.isclose(float(p.default), default, rel_tol=1e-7)
      assert same and type(p.default) is type(default), f'{label}({name}): default {p.default!r}, reference {default!r}'
  extra = names[len(ref_names):]
  assert extra == ALLOWED_EXTRA_ARGS.get(label, []) or set(extra) <= set(ALLOWED_EXTRA_ARGS.get(label, [])), f'{label}: undeclared extra arguments {extra}'
  for p in ps[len(ref_names):]:
    assert p.default is not inspect.Parameter.empty, f'{label}({p.name}): an extra argument needs a default'


@pytest.mark.parametrize('name', sorted(SURFACE['pybind']['functions']))
def test_function_signature(ext, name):
  assert hasattr(ext, name), f'{name} missing from torch_darktable_extension'
  _check_args(name, getattr(ext, name), SURFACE['pybind']['functions'][name])
  # the typing stub names the same function (its argument NAMES differ from the pybind ones for the codec's first
  # argument -- `image` / `packed_data` vs `input`; pybind decides what a keyword call accepts)
  assert name in SURFACE['pyi']['functions']


@pytest.mark.parametrize('name', sorted(SURFACE['pybind']['classes']))
def test_class_surface(ext, name):
  ref = SURFACE['pybind']['classes'][name]
  cls = getattr(ext, name)
  if ref['init']:
    _check_args(f'{name}.__init__', cls.__init__, ref['init'], skip_self=True, member_defaults=ref.get('member_defaults'))
  for m, args in ref['methods'].items():
    if m.startswith('__'):
      continue
    assert callable(getattr(cls, m, None)), f'{name}.{m} missing'
    if args:  # Jpeg.encode is registered as a lambda without py::arg names
      _check_args(f'{name}.{m}', getattr(cls, m), args, skip_self=True)
  for prop, mode in ref['properties'].items():
    attr = inspect.getattr_static(cls, prop, None)
    if isinstance(attr, property):
      assert (attr.fset is not None) == (mode == 'rw'), f'{name}.{prop}: reference is {mode}'
    else:  # plain attribute set in __init__ (TonemapParams' def_readwrite fields)
      assert mode == 'rw' and prop in _params(cls.__init__, True).__str__(), f'{name}.{prop} missing'
  public = {m for m, v in vars(cls).items() if callable(v) and not m.startswith('_')}
  extra = public - set(ref['methods'])
  assert extra <= ALLOWED_EXTRA_METHODS.get(name, set()), f'{name}: undeclared extra methods {extra - ALLOWED_EXTRA_METHODS.get(name, set())}'
  # the stub agrees on the attribute names
  if name in SURFACE['pyi']['classes']:
    for a in SURFACE['pyi']['classes'][name]['attributes']:
      assert hasattr(cls, a) or a in [p.name for p in _params(cls.__init__, True)], f'{name}.{a} (stub attribute) missing'


@pytest.mark.parametrize('name', sorted(SURFACE['pybind']['enums']))
def test_enum_members(ext, name):
  enum = getattr(ext, name)
  for member in SURFACE['pybind']['enums'][name]:
    assert hasattr(enum, member), f'{name}.{member} missing'
  if name.startswith('Jpeg'):  # .export_values(): members are also module attributes
    for member in SURFACE['pybind']['enums'][name]:
      assert hasattr(ext, member)


def test_no_undeclared_public_functions(ext):
  """What the module exports beyond the reference's registrations is a closed list (fused ops and helpers)."""
  ref = set(SURFACE['pybind']['functions']) | set(SURFACE['pybind']['classes']) | set(SURFACE['pybind']['enums'])
  ref |= {m for e in ('JpegInputFormat', 'JpegSubsampling') for m in SURFACE['pybind']['enums'][e]} | {'JpegException'}
  mine = {n for n, v in vars(ext).items() if not n.startswith('_') and (inspect.isfunction(v) or inspect.isclass(v)) and getattr(v, '__module__', '') == ext.__name__}
  extra = mine - ref
  assert extra <= ALLOWED_EXTRA_NAMES, f'undeclared public names: {sorted(extra - ALLOWED_EXTRA_NAMES)}'
