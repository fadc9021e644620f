"""In-tree build of the native libraries (gfx950 only).

  lib/libtsamd.so      hand-written HIP kernels behind the C-ABI of include/tsamd.h
                       (hipcc --offload-arch=gfx950, no torch dependency)
  lib/_tsamd_ops.so    torch operator glue: registers torch_sparse::* ops with the
                       reference's schemas on top of libtsamd.so (g++, links torch)

Run as ``python pytorch_sparse_amd/build.py`` or through ``__graft_entry__.build()``.
Objects are cached under ``build/`` and rebuilt when a source or header is newer.
hipcc cross-compiles, so this works on a machine without a GPU.
"""
import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, 'csrc')
INCLUDE = os.path.join(ROOT, 'include')
LIBDIR = os.path.join(PKG, 'lib')
OBJDIR = os.path.join(ROOT, 'build', 'obj')
ARCH = 'gfx950'

HIP_SOURCES = ['api.hip', 'spmm.hip', 'spmm_min.hip', 'spmm_max.hip', 'spmm_partial.hip', 'spmm_bw.hip', 'spmm_bw_list.hip', 'spmm_ref_order.hip', 'spmm_coo.hip', 'convert.hip', 'scan.hip', 'sort.hip',
               'coalesce.hip', 'spspmm.hip', 'select.hip', 'sample.hip', 'segreduce.hip']
OPS_SOURCES = ['ops_spmm.cpp', 'ops_storage.cpp', 'ops_sample.cpp']
# translation units that #include another .hip file (rebuilt when that one changes)
HIP_INCLUDES = {'spmm_partial.hip': ['spmm.hip'], 'spmm_min.hip': ['spmm.hip'], 'spmm_max.hip': ['spmm.hip']}


def _hipcc():
    exe = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(exe):
        raise RuntimeError('hipcc not found; a ROCm toolchain is required to build libtsamd.so')
    return exe


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _headers():
    hs = [os.path.join(INCLUDE, f) for f in os.listdir(INCLUDE) if f.endswith('.h')]
    hs += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    return hs


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError('command failed: %s\n%s' % (' '.join(cmd), r.stdout))
    return r.stdout


def build_kernels(verbose=True, force=False):
    os.makedirs(OBJDIR, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()
    headers = _headers()
    flags = ['-O3', '-std=c++17', '-fPIC', '--offload-arch=' + ARCH, '-I' + INCLUDE, '-I' + CSRC,
             '-Wall', '-Wno-unused-function']
    # extra -D / -m flags (tuning constants such as -DTSAMD_SPSPMM_LG_RANGE=14); a change of the flag
    # set rebuilds every object (the objects record nothing about the flags they were built with)
    flags += os.environ.get('TSAMD_HIPCC_FLAGS', '').split()
    stamp = os.path.join(OBJDIR, 'hipcc_flags.txt')
    old_flags = open(stamp).read() if os.path.exists(stamp) else None
    if old_flags != ' '.join(flags):
        force = force or old_flags is not None or bool(os.environ.get('TSAMD_HIPCC_FLAGS'))
        with open(stamp, 'w') as f:
            f.write(' '.join(flags))
    jobs = []
    objs = []
    for src in HIP_SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJDIR, src + '.o')
        objs.append(o)
        deps = [s] + headers + [os.path.join(CSRC, d) for d in HIP_INCLUDES.get(src, [])]
        if force or _newer(o, deps):
            jobs.append([hipcc] + flags + ['-c', s, '-o', o])
    if jobs:
        if verbose:
            print('[build] hipcc: %d translation unit(s) for %s' % (len(jobs), ARCH), flush=True)
        with cf.ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            for out in ex.map(_run, jobs):
                if out.strip() and verbose:
                    print(out)
    lib = os.path.join(LIBDIR, 'libtsamd.so')
    if force or _newer(lib, objs):
        _run([hipcc, '-shared', '-fPIC', '--offload-arch=' + ARCH, '-o', lib] + objs)
        if verbose:
            print('[build] linked', lib, flush=True)
    return lib


def build_ops(verbose=True, force=False, kernels_done=None):
    """torch_sparse::* operator library (host-only C++, compiled with g++).  kernels_done: a future of
    build_kernels() to wait for before linking (build_all compiles the two libraries side by side)."""
    import torch
    from torch.utils import cpp_extension as ce
    os.makedirs(OBJDIR, exist_ok=True)
    srcs = [os.path.join(CSRC, s) for s in OPS_SOURCES]
    if not all(os.path.exists(s) for s in srcs):
        return None
    lib = os.path.join(LIBDIR, '_tsamd_ops.so')
    headers = _headers()
    tlib = os.path.join(os.path.dirname(torch.__file__), 'lib')
    inc = ce.include_paths() + ['/opt/rocm/include', INCLUDE, CSRC]
    flags = ['-O2', '-std=c++17', '-fPIC', '-D__HIP_PLATFORM_AMD__=1', '-DUSE_ROCM=1',
             '-D_GLIBCXX_USE_CXX11_ABI=%d' % int(torch._C._GLIBCXX_USE_CXX11_ABI),
             '-Wno-deprecated-declarations']
    objs = []
    jobs = []
    for s in srcs:
        o = os.path.join(OBJDIR, os.path.basename(s) + '.o')
        objs.append(o)
        if force or _newer(o, [s] + headers):
            jobs.append(['g++'] + flags + ['-I' + i for i in inc] + ['-c', s, '-o', o])
    if jobs and verbose:
        print('[build] g++: torch operator glue (%d TU)' % len(jobs), flush=True)
    if jobs:
        with cf.ThreadPoolExecutor(max_workers=len(jobs)) as ex:
            for out in ex.map(_run, jobs):
                if out.strip() and verbose:
                    print(out)
    if kernels_done is not None:
        kernels_done.result()  # libtsamd.so is linked (or its build raised)
    if force or _newer(lib, objs + [os.path.join(LIBDIR, 'libtsamd.so')]):
        _run(['g++', '-shared', '-fPIC', '-o', lib] + objs +
             ['-L' + LIBDIR, '-ltsamd', '-L' + tlib, '-ltorch', '-ltorch_cpu', '-lc10',
              '-ltorch_hip', '-lc10_hip', '-Wl,-rpath,$ORIGIN', '-Wl,-rpath,' + tlib])
        if verbose:
            print('[build] linked', lib, flush=True)
    return lib


def build_all(verbose=True, force=False):
    """Both libraries.  The g++ translation units of the operator glue (~1 min of torch headers each) are compiled
    WHILE hipcc works on the kernels; only the glue's link step waits for libtsamd.so."""
    with cf.ThreadPoolExecutor(max_workers=2) as ex:
        fk = ex.submit(build_kernels, verbose, force)
        fo = ex.submit(build_ops, verbose, force, fk)
        return fk.result(), fo.result()


if __name__ == '__main__':
    build_all(force='--force' in sys.argv)
