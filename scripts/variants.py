"""Build A/B variants of libtsamd.so with -D overrides into build/variants/<name>.so."""
import os, subprocess, sys, concurrent.futures as cf
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, 'pytorch_sparse_amd', 'csrc')
OUT = os.path.join(ROOT, 'build', os.environ.get('TSAMD_VARIANT_DIR', 'ab'))
def build(name, defs, sources=('api.hip', 'spmm.hip', 'spmm_bw.hip', 'convert.hip')):
    os.makedirs(OUT, exist_ok=True)
    so = os.path.join(OUT, name + '.so')
    cmd = ['hipcc', '-O3', '-std=c++17', '-fPIC', '-shared', '--offload-arch=gfx950', '-I' + os.path.join(ROOT, 'include'), '-I' + CSRC]
    cmd += ['-D' + d for d in defs] + [os.path.join(CSRC, s) for s in sources] + ['-o', so]
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
