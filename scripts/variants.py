"""Build A/B variants of libtsamd.so with -D overrides into build/ab/<name>.so (build/variants/ is
gpurun-ignored).  Usage: python scripts/variants.py base: fast:TSAMD_UNROLL=8,TSAMD_WPB=2"""
import os, subprocess, sys, concurrent.futures as cf
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'pytorch_sparse_amd', 'csrc')
OUT = os.path.join(ROOT, 'build', os.environ.get('TSAMD_VARIANT_DIR', 'ab'))
def build(name, defs, sources=None):
    # every translation unit: a variant that is preloaded in front of the shipped library (LD_PRELOAD for the
    # torch-op path) or selected with TSAMD_LIB (ctypes path) must resolve all tsamd_* symbols by itself
    sources = sources or sorted(f for f in os.listdir(CSRC) if f.endswith('.hip'))
    os.makedirs(OUT, exist_ok=True)
    so = os.path.join(OUT, name + '.so')
    cmd = ['hipcc', '-O3', '-std=c++17', '-fPIC', '-shared', '--offload-arch=gfx950', '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC]
    # (variants are experiment builds: the TSAMD_* environment switches and the rejected kernel variants are compiled in)
    cmd += ['-DTSAMD_EXPERIMENTS=1'] + ['-D' + d for d in defs] + [os.path.join(CSRC, s) for s in sources] + ['-o', so]
    subprocess.check_call(cmd)
    return so
if __name__ == '__main__':
    specs = {}
    for a in sys.argv[1:]:
        name, _, d = a.partition(':')
        specs[name] = [x for x in d.split(',') if x]
    with cf.ThreadPoolExecutor(8) as ex:
        for so in ex.map(lambda kv: build(*kv), specs.items()):
            print(so)
