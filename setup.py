"""pip / setuptools entry: builds the two native libraries in-tree (hipcc for gfx950, g++ for the torch
operator glue -- pytorch_sparse_amd/build.py) and installs the package with them as package data.

    pip install --no-build-isolation -e .        # or: python setup.py build_ext --inplace

There is no CPU build: the libraries contain the only implementation."""
import importlib.util
import os

from setuptools import Command, find_packages, setup
from setuptools.command.build_py import build_py

ROOT = os.path.dirname(os.path.abspath(__file__))


def _build_native():
    spec = importlib.util.spec_from_file_location('_tsamd_build', os.path.join(ROOT, 'pytorch_sparse_amd', 'build.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.build_all(verbose=True)


class BuildNative(Command):
    description = 'compile libtsamd.so (HIP, gfx950) and _tsamd_ops.so (torch operator library) in-tree'
    user_options = [('inplace', 'i', 'ignored: the libraries are always built in-tree')]

    def initialize_options(self):
        self.inplace = 1

    def finalize_options(self):
        pass

    def run(self):
        _build_native()


class BuildPy(build_py):
    def run(self):
        _build_native()
        super().run()


setup(
    name='pytorch_sparse_amd',
    version='0.1.0',
    description='The sparse-matmul hot path of rusty1s/pytorch_sparse, native on AMD MI355X (gfx950)',
    packages=find_packages(include=['pytorch_sparse_amd', 'pytorch_sparse_amd.*']),
    package_data={'pytorch_sparse_amd': ['lib/*.so', 'csrc/*']},
    include_package_data=True,
    python_requires='>=3.9',
    install_requires=[],  # torch (ROCm build) is expected to be present already
    cmdclass={'build_ext': BuildNative, 'build_py': BuildPy},
    zip_safe=False,
)
