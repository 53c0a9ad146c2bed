"""Packaging: the native core is built in-tree (lightgbm_amd/build.py drives g++ and hipcc --offload-arch=gfx950)
and shipped as package data.

    python setup.py build_native          # just the shared libraries
    pip install --no-build-isolation .    # builds them, then installs the package
"""
import os
import sys

from setuptools import Command, find_packages, setup
from setuptools.command.build_py import build_py

HERE = os.path.dirname(os.path.abspath(__file__))


def _build_native(with_hip: bool):
    sys.path.insert(0, HERE)
    from lightgbm_amd import build as native
    native.build_host(verbose=True)
    if with_hip:
        if native.hipcc_path() is None:
            print("hipcc not found: building the CPU core only", file=sys.stderr)
        else:
            native.build_hip(verbose=True)
            native.build_hip(verbose=True, debug=True)


class BuildNative(Command):
    description = "compile _lga_host.so, _lga_hip.so and _lga_hip_debug.so next to the Python sources"
    user_options = [("cpu-only", None, "skip the HIP libraries")]

    def initialize_options(self):
        self.cpu_only = False

    def finalize_options(self):
        pass

    def run(self):
        _build_native(not self.cpu_only)


class BuildPyWithNative(build_py):
    def run(self):
        _build_native(os.environ.get("LGA_CPU_ONLY", "0") != "1")
        super().run()


setup(
    name="lightgbm_amd",
    version="0.1.0",
    description="MI355X-native gradient-boosted decision trees (LightGBM-style API, HIP kernels, RCCL)",
    packages=find_packages(include=["lightgbm_amd", "lightgbm_amd.*"]),
    package_data={"lightgbm_amd": ["*.so", "csrc/*.hpp", "csrc/*.cpp", "csrc/hip/*"]},
    python_requires=">=3.9",
    install_requires=["numpy"],
    extras_require={"gpu": ["torch"], "frames": ["pandas"], "sklearn": ["scikit-learn"]},
    cmdclass={"build_native": BuildNative, "build_py": BuildPyWithNative},
    entry_points={"console_scripts": ["lightgbm-amd=lightgbm_amd.cli:main"]},
)
