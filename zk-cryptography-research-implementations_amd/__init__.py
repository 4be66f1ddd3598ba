"""zk-cryptography-research-implementations_amd -- MI355X-native multilinear prover path.

Python host binding (ctypes) of libzkmle_amd.so, mirroring the reference's Rust items for the
path: `MultilinearPolynomial::{new, evaluate, partial_evaluate, ...}`
(polynomials/src/multilinear/evaluation_form.rs).  Tables live in HBM behind `zk_table` handles;
every operation runs hand-written HIP kernels through the C ABI of include/zkmle.h.

There is NO CPU fallback: if the shared library is missing, or no HIP device is usable, the
operations raise.  (The directory name contains '-', so import it with
`__graft_entry__.import_package()`, which registers it as `zkmle_amd`.)
"""
from . import _lib
from ._lib import FR381, FQ381, BN254_FQ, BN254_FR, ZkError, ReferencePanic, lib, library_path  # noqa: F401
from .mle import MultilinearPolynomial, from_ints, to_ints, limbs  # noqa: F401
from . import sumcheck  # noqa: F401
from .sumcheck import Transcript, ProductPolynomial, SumPolynomial, Prover, Verifier  # noqa: F401
from . import gkr  # noqa: F401
from .gkr import Circuit, Gate, Layer, Operator  # noqa: F401
from . import kzg  # noqa: F401
from .kzg import G1Bases, TrustedSetup, MultilinearKZG, MultilinearKZGProof  # noqa: F401
from . import sharded  # noqa: F401

__all__ = ["MultilinearPolynomial", "FR381", "FQ381", "BN254_FQ", "BN254_FR", "ZkError", "ReferencePanic",
           "from_ints", "to_ints", "limbs", "lib", "library_path"]
