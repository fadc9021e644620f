"""CPU oracle for the sparse-matmul hot path -- TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this package.  The product (``pytorch_sparse_amd``) never
does: it has no CPU compute path and fails loudly without its HIP library.

Two checkers live here:

* ``oracle.c_oracle``   -- ``ts_oracle.c``, a scalar C restatement of the reference
  algorithms (built with gcc into ``oracle/libts_oracle.so``), driven through numpy;
* ``oracle.ref``        -- the reference's own CPU kernels compiled unmodified from
  ``/root/reference/csrc`` into ``oracle/_ref/`` (see ``build_ref.py``), when present.

``oracle.np_oracle`` restates the Python-level pipelines (sort, coalesce, transpose,
SpSpMM) with numpy.
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, 'libts_oracle.so')
SRC = os.path.join(HERE, 'ts_oracle.c')


def build(verbose=False):
    """gcc -O2 -shared ts_oracle.c -> libts_oracle.so (rebuilt when the source is newer)."""
    if os.path.exists(LIB) and os.path.getmtime(LIB) >= os.path.getmtime(SRC):
        return LIB
    cmd = ['gcc', '-O2', '-fPIC', '-shared', '-std=c99', '-o', LIB, SRC, '-lm']
    if verbose:
        print('[oracle]', ' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB
