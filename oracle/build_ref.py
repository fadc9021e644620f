"""Build the reference's own CPU hot path into oracle/_ref/ (TEST INFRASTRUCTURE ONLY).

Compiles, unmodified and where they lie under /root/reference:
    csrc/spmm.cpp  csrc/cpu/spmm_cpu.cpp  csrc/convert.cpp  csrc/cpu/convert_cpu.cpp
    csrc/{sample,rw,saint,relabel,diag,neighbor_sample}.cpp + their csrc/cpu/*_cpu.cpp   (SURVEY 8f widening)
with g++ (mirrors setup.py:67-81 of the reference: -O3 -fopenmp -DAT_PARALLEL_OPENMP,
no WITH_CUDA => CPU only) into ``oracle/_ref/libts_ref.so``.  The two op files are
included through ``ref_wrap_*.cpp`` so their registrations land in ``ts_ref::`` instead
of ``torch_sparse::`` (see ref_wrap.h); ``oracle/shim`` supplies the missing
parallel-hashmap header.  Nothing is copied into the repository; the .so is git-ignored
but travels to the GPU box with the snapshot.

Usage:  python oracle/build_ref.py        (no-op when /root/reference is absent)
"""
import concurrent.futures as cf
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get('TS_REFERENCE', '/root/reference')
OUT = os.path.join(HERE, '_ref')
LIB = os.path.join(OUT, 'libts_ref.so')


def build(verbose=True):
    csrc = os.path.join(REF, 'csrc')
    if not os.path.isdir(csrc):
        if verbose:
            print('[oracle/_ref] %s not present: keeping prebuilt %s' % (REF, LIB))
        return LIB if os.path.exists(LIB) else None
    import torch
    from torch.utils import cpp_extension as ce
    os.makedirs(OUT, exist_ok=True)
    families = ['spmm', 'convert', 'sample', 'rw', 'saint', 'relabel', 'diag', 'neighbor_sample']
    srcs, deps = [], [os.path.join(HERE, 'ref_wrap.h'), os.path.join(csrc, 'cpu', 'reducer.h'),
                      os.path.abspath(__file__)]
    for fam in families:
        srcs += [os.path.join(HERE, 'ref_wrap_%s.cpp' % fam), os.path.join(csrc, 'cpu', fam + '_cpu.cpp')]
        deps.append(os.path.join(csrc, fam + '.cpp'))
    deps += srcs
    if os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in deps):
        return LIB
    tlib = os.path.join(os.path.dirname(torch.__file__), 'lib')
    inc = [csrc, HERE, os.path.join(HERE, 'shim')] + ce.include_paths()
    flags = ['-O3', '-fopenmp', '-DAT_PARALLEL_OPENMP', '-Wno-sign-compare', '-std=c++17', '-fPIC',
             '-D_GLIBCXX_USE_CXX11_ABI=%d' % int(torch._C._GLIBCXX_USE_CXX11_ABI)]
    objs = [os.path.join(OUT, os.path.basename(s) + '.o') for s in srcs]

    def cc(so):
        s, o = so
        subprocess.check_call(['g++'] + flags + ['-I' + i for i in inc] + ['-c', s, '-o', o])
    if verbose:
        print('[oracle/_ref] compiling the reference CPU path from', csrc, flush=True)
    with cf.ThreadPoolExecutor(8) as ex:
        list(ex.map(cc, zip(srcs, objs)))
    subprocess.check_call(['g++', '-shared', '-fopenmp', '-o', LIB] + objs +
                          ['-L' + tlib, '-ltorch', '-ltorch_cpu', '-lc10', '-Wl,-rpath,' + tlib])
    for o in objs:
        os.remove(o)
    if verbose:
        print('[oracle/_ref] built', LIB, flush=True)
    return LIB


if __name__ == '__main__':
    build()
