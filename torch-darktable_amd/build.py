"""Builds libtdk_hip.so (gfx950) in-tree with hipcc: one object per .hip file, in parallel.

  python torch-darktable_amd/build.py [--force] [--jobs N]

The library lands next to the Python package (torch_darktable/libtdk_hip.so) so it travels
with the source tree; no JIT cache, no torch headers.
"""

from __future__ import annotations

import argparse
import fcntl
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

HERE = Path(__file__).resolve().parent
CSRC = HERE / 'csrc'
OBJ = HERE / 'build'
LIB = HERE / 'torch_darktable' / 'libtdk_hip.so'
STAMP = HERE / 'torch_darktable' / 'libtdk_hip.so.sha256'  # hash of the sources + flags the library was built from

HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
ARCH = 'gfx950'
# -ffp-contract=off: no FMA contraction, so + - * / kernels match the strict-fp32 oracle bit for bit.
# -fno-slp-vectorize: packed fp32 VALU (v_pk_*) issues at half rate on gfx950 and the packing moves
# cost more than they save (Wiener tile kernel: 4965 -> 4369 VALU instructions, 228 -> 151 VGPRs).
CXXFLAGS = ['-O3', '-std=c++17', f'--offload-arch={ARCH}', '-ffp-contract=off', '-fno-slp-vectorize', '-fPIC', '-fvisibility=hidden',
            '-Wall', '-Wno-unused-function']


CXXFLAGS += os.environ.get('TDK_EXTRA_FLAGS', '').split()  # experiments: e.g. TDK_EXTRA_FLAGS=-DTDK_WIENER_SHARED_ACC=1


def _inputs():
  return sorted(CSRC.glob('*.hip')) + sorted(CSRC.glob('*.h')) + [HERE.parent / 'include' / 'tdk_hip.h']


def source_hash() -> str:
  h = hashlib.sha256(' '.join(CXXFLAGS).encode())
  for f in _inputs():
    h.update(f.name.encode())
    h.update(f.read_bytes())
  return h.hexdigest()


def is_current() -> bool:
  """True when the in-tree library was built from exactly these sources and flags.  Content hash,
  not mtimes: a copied tree (the GPU box gets a snapshot) keeps no trustworthy timestamps."""
  return LIB.exists() and STAMP.exists() and STAMP.read_text().strip() == source_hash()


def ensure_current(verbose: bool = False) -> Path:
  """Build only if the library does not match the sources.  Safe to call from several processes
  at once (one rank per GPU): an exclusive file lock serialises the builders and the losers find
  the finished library when they get the lock."""
  if is_current():
    return LIB
  OBJ.mkdir(exist_ok=True)
  with open(OBJ / '.lock', 'w') as lock:
    fcntl.flock(lock, fcntl.LOCK_EX)
    try:
      if not is_current():
        build(force=True, verbose=verbose)  # object mtimes cannot be trusted either
    finally:
      fcntl.flock(lock, fcntl.LOCK_UN)
  return LIB


def _stale(target: Path, deps) -> bool:
  return (not target.exists()) or any(d.stat().st_mtime > target.stat().st_mtime for d in deps)


def build(force: bool = False, jobs: int | None = None, verbose: bool = False) -> Path:
  OBJ.mkdir(exist_ok=True)
  headers = list(CSRC.glob('*.h')) + [HERE.parent / 'include' / 'tdk_hip.h']
  sources = sorted(CSRC.glob('*.hip'))
  todo = []
  objs = []
  for src in sources:
    obj = OBJ / (src.stem + '.o')
    objs.append(obj)
    if force or _stale(obj, [src, *headers]):
      todo.append((src, obj))

  def compile_one(job):
    src, obj = job
    cmd = [HIPCC, *CXXFLAGS, '-c', str(src), '-o', str(obj)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
      raise RuntimeError(f'hipcc failed for {src.name}:\n{r.stderr}')
    if verbose and r.stderr.strip():
      print(r.stderr, file=sys.stderr)
    return src.name

  if todo:
    with ThreadPoolExecutor(max_workers=jobs or min(8, os.cpu_count() or 1)) as ex:
      for name in ex.map(compile_one, todo):
        if verbose:
          print(f'  compiled {name}')
  if todo or force or _stale(LIB, objs):
    tmp = LIB.with_suffix(f'.so.tmp{os.getpid()}')
    cmd = [HIPCC, '-shared', '-fPIC', f'--offload-arch={ARCH}', '-o', str(tmp), *map(str, objs)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
      raise RuntimeError(f'link failed:\n{r.stderr}')
    os.replace(tmp, LIB)  # atomic: a process that is loading the old library keeps its mapping
  STAMP.write_text(source_hash() + '\n')
  return LIB


if __name__ == '__main__':
  ap = argparse.ArgumentParser()
  ap.add_argument('--force', action='store_true')
  ap.add_argument('--jobs', type=int, default=None)
  args = ap.parse_args()
  print(build(force=args.force, jobs=args.jobs, verbose=True))
