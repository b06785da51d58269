"""Build variants/exp.so: every .hip compiled with -DTDK_EXPERIMENTS (the environment knobs of the experiments: TDK_RCD_QUAD,
TDK_RCD_LDS_PAD, TDK_WIENER_LDS_PAD, TDK_BIL_LDS_PAD, TDK_WIENER_TR, ...).  Loaded through TDK_LIB_PATH; the in-tree library is untouched.
    python profiles/build_experiments.py [extra flags]"""
import importlib.util
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
PKG = ROOT / 'torch-darktable_amd'


def main():
    spec = importlib.util.spec_from_file_location('tdk_build', PKG / 'build.py')
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    out = ROOT / 'variants'
    out.mkdir(exist_ok=True)
    tmp = out / 'exp_obj'
    tmp.mkdir(exist_ok=True)
    srcs = sorted((PKG / 'csrc').glob('*.hip'))

    def one(src):
        obj = tmp / (src.stem + '.o')
        subprocess.run([b.HIPCC, *b.CXXFLAGS, '-DTDK_EXPERIMENTS', *sys.argv[1:], f'-I{PKG / "csrc"}', f'-I{ROOT / "include"}', '-c', str(src), '-o', str(obj)], check=True)
        return str(obj)

    with ThreadPoolExecutor(4) as ex:
        objs = list(ex.map(one, srcs))
    lib = out / 'exp.so'
    subprocess.run([b.HIPCC, '-shared', '-fPIC', f'--offload-arch={b.ARCH}', '-o', str(lib), *objs], check=True)
    print(lib)


if __name__ == '__main__':
    main()
