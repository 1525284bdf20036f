"""fermiflow_amd -- MI355X-native VMC inner loop behind FermiFlow's Python interface.

Module names follow the reference's flat `src/` layout (orbitals, slater, base_dist, MLP, equivariant_funs,
flow, potentials, utils, VMC, NeuralODE.nnModule).  Everything numerical is in libfermiflow_hip.so
(fermiflow_amd/csrc, C ABI in include/fermiflow.h); there is no CPU implementation in this package.
"""
import torch

torch.set_default_dtype(torch.float64)   # the reference does this at the top of every module (e.g. src/VMC.py:2)

from .orbitals import HO2D, HO3D, Orbitals, Orbital, Orbital3D          # noqa: E402,F401
from .slater import LogAbsSlaterDet, LogAbsSlaterDetMultStates, logabsslaterdet, logabsslaterdetmultstates  # noqa: E402,F401
from .base_dist import FreeFermion                      # noqa: E402,F401
from .MLP import MLP                                    # noqa: E402,F401
from .equivariant_funs import Backflow                  # noqa: E402,F401
from .flow import CNF                                   # noqa: E402,F401
from .potentials import HO, CoulombPairPotential        # noqa: E402,F401
from .VMC import GSVMC, BetaVMC                         # noqa: E402,F401
from .utils import y_grad_laplacian                     # noqa: E402,F401
from .NeuralODE.nnModule import solve_ivp_nnmodule      # noqa: E402,F401
from . import checkpoint                                # noqa: E402,F401
