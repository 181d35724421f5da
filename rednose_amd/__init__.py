"""rednose_amd -- MI355X-native batched EKF predict/update engine behind the rednose filter API.

Only the hot path of commaai/rednose is provided: sympy model definitions go through
`rednose_amd.helpers.ekf_sym.gen_code`, which emits hand-structured HIP kernels for gfx950 and a C-ABI
shared library; `EKF_sym` / `BatchedEKF` drive it from Python.  There is no CPU fallback: every
compute entry point launches on the GPU and raises if no HIP device / library is available.
"""

__version__ = "0.1.0"
