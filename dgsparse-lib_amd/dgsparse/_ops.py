"""Loads the ``torch.ops.dgsparse_spmm.*`` registry: the C++ ``TORCH_LIBRARY`` in ``_spmm_hip*.so`` next to the package
(csrc/torch_binding.cpp), found and loaded the way the reference loads ``_spmm_cuda*.so`` (dgsparse/__init__.py:16-26).

It mirrors the reference binding src/spmm.cpp:36-270: ``spmm_sum / spmm_max / spmm_min / spmm_mean`` with the positional
schema ``(rowptr, col, values, colptr, row, csr2csc, dense, has_value, algorithm)``, ``csr2csc``, and four
``torch::autograd::Function`` classes whose backward is one SDDMM (grad of the sparse values) + one SpMM on the CSC arrays
(grad of the dense operand).  Additions: ``sddmm``, ``csr2csc_perm``, ``spmm_raw``, and the ``*_p`` variants of the four
operators that take what a ``Storage`` keeps next to its CSC view (permuted values, forward / backward locality plans).

There is exactly one binding (an earlier Python duplicate of it was removed): without the ``.so`` the import fails.
"""
import importlib.machinery
import os

import torch

_spec = importlib.machinery.PathFinder().find_spec('_spmm_hip', [os.path.dirname(__file__)])
if _spec is None:
    raise ImportError(f"Could not find module '_spmm_hip' in {os.path.dirname(__file__)}. Build it with "
                      f"`make -C dgsparse-lib_amd/csrc` (or `python -c 'import __graft_entry__ as g; g.build()'`).")
torch.ops.load_library(_spec.origin)
NATIVE_BINDING = True
