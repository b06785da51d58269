"""Generates tests/golden/wrapper_surface.json: the signatures of the reference's Python wrappers (the L3 API that
BASELINE.json's north_star says stays unchanged), parsed as TEXT with `ast` (no import, nothing executed) from

  /root/reference/torch_darktable/{bayer,debayer,denoise,local_contrast,tonemap,color_conversion,white_balance,jpeg}.py
  /root/reference/torch_darktable/__init__.py      the exported names (__all__ or the import list)

Per module: module-level functions and classes; per function / method: the argument names in order, which are
keyword-only, and the defaults as source text; per class: its base names and (for dataclasses) its annotated fields.  The JSON is
data; tests/test_wrapper_surface.py parses this repo's wrappers the same way and compares.  Re-run only in the build
container (the reference tree does not exist on the GPU box):

  python tests/golden/make_wrapper_surface.py
"""

from __future__ import annotations

import ast
import json
import sys
from pathlib import Path

MODULES = ['bayer', 'debayer', 'denoise', 'local_contrast', 'tonemap', 'color_conversion', 'white_balance', 'jpeg']


def signature(fn: ast.FunctionDef | ast.AsyncFunctionDef) -> dict:
  a = fn.args
  pos = [x.arg for x in a.posonlyargs + a.args]
  defaults = [None] * (len(pos) - len(a.defaults)) + [ast.unparse(d) for d in a.defaults]
  kwonly = [x.arg for x in a.kwonlyargs]
  kwdefaults = [ast.unparse(d) if d is not None else None for d in a.kw_defaults]
  return {'args': pos, 'defaults': defaults, 'kwonly': kwonly, 'kwdefaults': kwdefaults,
          'vararg': a.vararg.arg if a.vararg else None, 'kwarg': a.kwarg.arg if a.kwarg else None,
          'decorators': sorted(ast.unparse(d).split('(')[0].split('.')[-1] for d in fn.decorator_list if ast.unparse(d).split('(')[0].split('.')[-1] in ('staticmethod', 'classmethod', 'property'))}


def module_surface(path: Path) -> dict:
  tree = ast.parse(path.read_text())
  out = {'functions': {}, 'classes': {}}
  for node in tree.body:
    if isinstance(node, (ast.FunctionDef, ast.AsyncFunctionDef)) and not node.name.startswith('_'):
      out['functions'][node.name] = signature(node)
    elif isinstance(node, ast.ClassDef) and not node.name.startswith('_'):
      cls = {'bases': [ast.unparse(b) for b in node.bases], 'methods': {}, 'fields': {}}
      for item in node.body:
        if isinstance(item, (ast.FunctionDef, ast.AsyncFunctionDef)) and (not item.name.startswith('_') or item.name == '__init__'):
          cls['methods'][item.name] = signature(item)
        elif isinstance(item, ast.AnnAssign) and isinstance(item.target, ast.Name):
          cls['fields'][item.target.id] = ast.unparse(item.value) if item.value is not None else None
        elif isinstance(item, ast.Assign) and len(item.targets) == 1 and isinstance(item.targets[0], ast.Name) and not item.targets[0].id.startswith('_'):
          cls['fields'][item.targets[0].id] = ast.unparse(item.value)
      out['classes'][node.name] = cls
  return out


def exported_names(init: Path) -> list[str]:
  tree = ast.parse(init.read_text())
  names: list[str] = []
  for node in tree.body:
    if isinstance(node, ast.Assign) and any(isinstance(t, ast.Name) and t.id == '__all__' for t in node.targets):
      return sorted(ast.literal_eval(node.value))
    if isinstance(node, ast.ImportFrom) and node.level >= 1:
      names += [a.asname or a.name for a in node.names]
  return sorted(set(names))


def surface(pkg: Path) -> dict:
  return {'modules': {m: module_surface(pkg / f'{m}.py') for m in MODULES if (pkg / f'{m}.py').exists()}, 'exports': exported_names(pkg / '__init__.py')}


if __name__ == '__main__':
  ref = Path(sys.argv[1]) if len(sys.argv) > 1 else Path('/root/reference/torch_darktable')
  out = Path(__file__).resolve().parent / 'wrapper_surface.json'
  s = surface(ref)
  s['_source'] = 'reference torch_darktable/*.py parsed with ast by tests/golden/make_wrapper_surface.py (names, argument order, defaults as text)'
  out.write_text(json.dumps(s, indent=1, sort_keys=True) + '\n')
  print(out, {m: (len(v['functions']), len(v['classes'])) for m, v in s['modules'].items()}, len(s['exports']), 'exports')
