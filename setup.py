"""Build hook of the `aqlm` distribution (MI355X implementation): `pip install .` / `python setup.py build_py` compile the
two native libraries next to the Python sources before they are collected as package data --

  aqlm_amd/libaqlm_hip.so   hipcc --offload-arch=gfx950 (aqlm_amd/csrc/Makefile; hipcc cross-compiles without a GPU)
  aqlm_amd/libaqlm_cpu.so   g++ -fopenmp            (aqlm_amd/csrc_cpu/Makefile)
  aqlm_amd/_aqlm_front.so   g++ + libtorch + pybind11 (aqlm_amd/csrc_front/Makefile; host glue of the decode call)

The reference ships pure Python and JIT-compiles its CUDA source at first use (inference_lib/setup.cfg:1-53,
cuda_kernel.py:7-11); ahead-of-time libraries are what this package loads, so they are built here.  Set AQLM_SKIP_HIP_BUILD=1 to package the CPU library only (hosts without ROCm)."""
import os
import subprocess
import sys

from setuptools import setup
from setuptools.command.build_py import build_py

ROOT = os.path.dirname(os.path.abspath(__file__))


def build_native():
    jobs = str(min(8, os.cpu_count() or 1))
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "aqlm_amd", "csrc_cpu")])
    if os.environ.get("AQLM_SKIP_HIP_BUILD") != "1":
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "aqlm_amd", "csrc"), "-j", jobs])
        # compiled host glue (needs torch): built against the interpreter that is installing the package, not whatever
        # `python3` is first on PATH (venv / conda installs), with the ROCm headers of this machine
        rocm = os.environ.get("ROCM_PATH") or os.environ.get("ROCM_HOME") or "/opt/rocm"
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "aqlm_amd", "csrc_front"), f"PYTHON={sys.executable}", f"ROCM_PATH={rocm}"])


class BuildPyWithNative(build_py):
    def run(self):
        build_native()
        super().run()


setup(
    name="aqlm",
    version="1.1.7",
    description="MI355X-native drop-in for the AQLM QuantizedLinear inference path (HIP kernels for gfx950 + native CPU kernels)",
    long_description=open(os.path.join(ROOT, "README.md")).read(),
    long_description_content_type="text/markdown",
    python_requires=">=3.8",
    install_requires=["torch>=2.2.0"],
    extras_require={"hf": ["transformers>=4.38.0", "accelerate>=0.27.0"], "dev": ["pytest", "numpy"]},
    packages=["aqlm", "aqlm_amd", "aqlm_amd.inference_kernels"],
    package_data={"aqlm_amd": ["libaqlm_hip.so", "libaqlm_cpu.so", "_aqlm_front.so"]},
    cmdclass={"build_py": BuildPyWithNative},
)
