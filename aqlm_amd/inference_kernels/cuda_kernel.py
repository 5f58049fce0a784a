"""Import-path compatibility: ``aqlm.inference_kernels.cuda_kernel.CUDA_KERNEL`` is how the reference's benchmark
reaches the raw kernels (benchmark/matmul_benchmark.py:6).  On MI355X the same names resolve to the HIP library."""
from .hip_kernel import CUDA_FOLDER, CUDA_KERNEL, HIP_FOLDER, HIP_KERNEL  # noqa: F401
