from .kernel_selector import get_backward_pass_kernel, get_forward_pass_kernel, optimize_for_training

__all__ = ["get_backward_pass_kernel", "get_forward_pass_kernel", "optimize_for_training"]
