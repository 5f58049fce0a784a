"""Drop-in alias: ``import aqlm`` resolves to the MI355X implementation in ``aqlm_amd``.

Hugging Face's integration does ``from aqlm import QuantizedLinear`` (transformers/integrations/aqlm.py:40) and
probes ``importlib.metadata.version("aqlm")`` (quantizer_aqlm.py:63-72; satisfied by the ``aqlm-*.dist-info``
directory shipped next to this package).  Every sub-module path of the reference keeps working --
``aqlm.inference``, ``aqlm.utils``, ``aqlm.inference_kernels.kernel_selector``,
``aqlm.inference_kernels.cuda_kernel`` ... -- because ``aqlm.<x>`` is imported as the very same module object as
``aqlm_amd.<x>`` (a meta-path alias, so nothing is executed twice and no op is registered twice).
"""
import importlib
import importlib.abc
import importlib.machinery
import importlib.util
import sys

import aqlm_amd as _impl


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    _prefix = __name__ + "."

    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith(self._prefix):
            return None
        real = _impl.__name__ + fullname[len(__name__):]
        try:
            if importlib.util.find_spec(real) is None:
                return None
        except (ImportError, ValueError):
            return None
        return importlib.machinery.ModuleSpec(fullname, self)

    def create_module(self, spec):
        return importlib.import_module(_impl.__name__ + spec.name[len(__name__):])

    def exec_module(self, module):
        pass


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())

from aqlm_amd import QuantizedLinear, __version__, optimize_for_training  # noqa: E402,F401
from aqlm_amd import get_backward_pass_kernel, get_forward_pass_kernel  # noqa: E402,F401
from aqlm_amd import SharedInputGroup, fuse_shared_input_linears, unfuse_shared_input_linears  # noqa: E402,F401

inference = importlib.import_module(__name__ + ".inference")
utils = importlib.import_module(__name__ + ".utils")
inference_kernels = importlib.import_module(__name__ + ".inference_kernels")
checkpoint = importlib.import_module(__name__ + ".checkpoint")
fusion = importlib.import_module(__name__ + ".fusion")
