"""Generates tests/golden/extension_surface.json: the Python-visible surface of the reference's compiled module
`torch_darktable.torch_darktable_extension`, parsed as TEXT (no import, nothing executed) from

  /root/reference/torch_darktable/csrc/extension.cpp:50-248   the pybind11 registrations: functions, classes,
                                                               argument names, defaults, properties, enum members
  /root/reference/torch_darktable/torch_darktable_extension.pyi  the typing stub: names only (its defaults are `...`)
  /root/reference/torch_darktable/csrc/tonemap/tonemap.h:6-15    TonemapParams' member initialisers (it is default-constructible)

The JSON is data (names, argument lists, default values); tests/test_extension_surface.py compares
`inspect.signature` of this repo's extension module with it.  Re-run only in the build container (the reference tree
does not exist on the GPU box):

  python tests/golden/make_extension_surface.py
"""

from __future__ import annotations

import ast
import json
import re
from pathlib import Path

REF = Path('/root/reference/torch_darktable')
OUT = Path(__file__).resolve().parent / 'extension_surface.json'


def _default(text: str):
  t = text.strip()
  if t in ('true', 'false'):
    return t == 'true'
  m = re.fullmatch(r'([-+0-9.eE]+)f?', t)
  if not m:
    raise ValueError(f'unparsed default {text!r}')
  v = m.group(1)
  return int(v) if re.fullmatch(r'[-+]?\d+', v) else float(v)


def _args(call: str):
  """[[name, default-or-None], ...] of the py::arg(...) entries of one registration call."""
  out = []
  for m in re.finditer(r'py::arg\("(\w+)"\)(\s*=\s*([^,()]+))?', call):
    out.append([m.group(1), _default(m.group(3)) if m.group(2) else None])
  return out


def _calls(stmt: str):
  """(method, text-inside-the-parentheses) for every `.def*(...)` / `.value(...)` of one statement, parentheses matched."""
  res = []
  for m in re.finditer(r'\.(def_property_readonly|def_property|def_readwrite|def|value)\s*\(', stmt):
    depth, i = 1, m.end()
    while depth:
      depth += {'(': 1, ')': -1}.get(stmt[i], 0)
      i += 1
    res.append((m.group(1), stmt[m.end():i - 1]))
  return res


def _statements(body: str):
  """Top-level statements of the module body: split at `;` outside parentheses and braces (the Jpeg class registers a lambda)."""
  out, depth, start = [], 0, 0
  for i, ch in enumerate(body):
    if ch in '({':
      depth += 1
    elif ch in ')}':
      depth -= 1
      if depth < 0:
        break
    elif ch == ';' and depth == 0:
      out.append(body[start:i])
      start = i + 1
  return out


def parse_cpp(src: str) -> dict:
  body = src[src.index('PYBIND11_MODULE'):]
  body = re.sub(r'//[^\n]*', '', body)
  surface = {'functions': {}, 'classes': {}, 'enums': {}}
  for stmt in _statements(body[body.index('{') + 1:]):
    m = re.search(r'py::class_<[^(]*\(m,\s*"(\w+)"\)', stmt)
    if m:
      cls = {'init': None, 'methods': {}, 'properties': {}}
      for kind, inner in _calls(stmt[m.end():]):
        if kind == 'def':
          if inner.lstrip().startswith('py::init'):
            a = _args(inner)
            if not a and re.match(r'\s*py::init<\s*>', inner):
              cls['default_constructible'] = True  # py::init<>(): the struct's member initialisers are the defaults
            if cls['init'] is None or len(a) > len(cls['init']):
              cls['init'] = a
          else:
            name = re.match(r'\s*"(\w+)"', inner).group(1)
            cls['methods'][name] = _args(inner)
        elif kind in ('def_property', 'def_readwrite'):
          cls['properties'][re.match(r'\s*"(\w+)"', inner).group(1)] = 'rw'
        elif kind == 'def_property_readonly':
          cls['properties'][re.match(r'\s*"(\w+)"', inner).group(1)] = 'ro'
      surface['classes'][m.group(1)] = cls
      continue
    m = re.search(r'py::enum_<[^(]*\(m,\s*"(\w+)"\)', stmt)
    if m:
      surface['enums'][m.group(1)] = [re.match(r'\s*"(\w+)"', inner).group(1) for kind, inner in _calls(stmt[m.end():]) if kind == 'value']
      continue
    m = re.search(r'\bm\.def\(\s*"(\w+)"', stmt)
    if m:
      surface['functions'][m.group(1)] = _args(stmt[m.end():])
  return surface


def parse_pyi(src: str) -> dict:
  tree = ast.parse(src)
  out = {'functions': {}, 'classes': {}}
  for node in tree.body:
    if isinstance(node, ast.FunctionDef):
      out['functions'][node.name] = [a.arg for a in node.args.args]
    elif isinstance(node, ast.ClassDef):
      c = {'methods': {}, 'attributes': []}
      for item in node.body:
        if isinstance(item, ast.FunctionDef):
          c['methods'][item.name] = [a.arg for a in item.args.args if a.arg != 'self']
        elif isinstance(item, ast.AnnAssign) and isinstance(item.target, ast.Name):
          c['attributes'].append(item.target.id)
      out['classes'][node.name] = c
  return out


def main():
  cpp = parse_cpp((REF / 'csrc' / 'extension.cpp').read_text())
  pyi = parse_pyi((REF / 'torch_darktable_extension.pyi').read_text())
  # TonemapParams is default-constructible: its member initialisers (csrc/tonemap/tonemap.h:7-10) are the defaults
  members = re.findall(r'float\s+(\w+)\s*=\s*([-+0-9.eE]+)f?\s*;', (REF / 'csrc' / 'tonemap' / 'tonemap.h').read_text())
  cpp['classes']['TonemapParams']['member_defaults'] = {n: float(v) for n, v in members}
  assert cpp['classes']['TonemapParams'].get('default_constructible') and len(members) == 4
  assert len(cpp["functions"]) == 27 and len(cpp["classes"]) == 8 and len(cpp['enums']) == 3, 'parser lost part of the module'
  doc = {
    'source': ['torch_darktable/csrc/extension.cpp:50-248', 'torch_darktable/torch_darktable_extension.pyi'],
    'pybind': cpp,
    'pyi': pyi,
  }
  OUT.write_text(json.dumps(doc, indent=1, sort_keys=True) + '\n')
  print(f'{OUT}: {len(cpp["functions"])} functions, {len(cpp["classes"])} classes, {len(cpp["enums"])} enums')


if __name__ == '__main__':
  main()
