"""Build a variant library: one .hip recompiled (other source file and / or extra flags), the other objects taken from the default build.
    python profiles/build_variant.py <stem> <out.so> [--src path.hip] [flags...]
e.g. python profiles/build_variant.py bilateral variants/bil_timing.so -DTDK_EXPERIMENTS -DTDK_BIL_TIMING=1"""
import importlib.util
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
PKG = ROOT / 'torch-darktable_amd'


def main():
    stem, out = sys.argv[1], Path(sys.argv[2])
    rest = sys.argv[3:]
    src = PKG / 'csrc' / f'{stem}.hip'
    if rest[:1] == ['--src']:
        src = Path(rest[1])
        rest = rest[2:]
    spec = importlib.util.spec_from_file_location('tdk_build', PKG / 'build.py')
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    b.build()
    out.parent.mkdir(exist_ok=True)
    others = [str(o) for o in (PKG / 'build').glob('*.o') if o.stem != stem]
    obj = out.with_suffix('.o')
    subprocess.run([b.HIPCC, *b.CXXFLAGS, f'-I{PKG / "csrc"}', *rest, '-c', str(src), '-o', str(obj)], check=True)
    subprocess.run([b.HIPCC, '-shared', '-fPIC', f'--offload-arch={b.ARCH}', '-o', str(out), *others, str(obj)], check=True)
    obj.unlink()
    print(out)


if __name__ == '__main__':
    main()
