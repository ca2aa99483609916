"""stochastic-inventory_amd -- MI355X-native engine for the finite-horizon SDP recursion of
RobinChen121/Stochastic-Inventory's src/sdp (hand-written HIP behind the C ABI of include/sdpgpu.h).

The directory name carries a hyphen (it is fixed by the build contract); import it as
`stochastic_inventory_amd` through the alias module at the repository root.
"""
from . import _abi
from ._abi import (FAMILY_BACKORDER, FAMILY_CASH, FAMILY_CASH_LEADTIME, FAMILY_LEADTIME, FAMILY_OVERDRAFT,
                   FAMILY_STAFF, FAMILY_SURVIVAL, KERNEL_AUTO, KERNEL_GATHER, KERNEL_WINDOW, SdpgpuDesc, SdpgpuError, SdpgpuStats, desc_defaults)
from .engine import SdpEngine
from .functors import (BackorderFunctor, CashFunctor, CashXRFunctor, CashLeadtimeFunctor, CustomFunctor, LeadtimeFunctor, OverdraftFunctor,
                       SurvivalFunctor, java_round)
from .multiitem import (Actions, CashRecursionMulti, CashRecursionMultiLead, CashRecursionMultiXR, CashStateMulti,
                        CashStateMultiLead, CashStateMultiXR, MultiLeadResult, multicash_solve, multilead_solve, multixr_solve)
from .pmf import BinomialDist, DiscreteDistribution, GammaDist, GetPmf, NormalDist, PoissonDist, UniformIntDist, staff_level_pmf
from .recursion import CLSP, CashLeadtimeRecursion, CashRecursion, CashRecursionXR, LeadtimeRecursion, Recursion, RiskRecursion
from .simulation import RiskSimulation, Sampling, Simulation
from .workforce import StaffFunctor, StaffRecursion, StaffState
from .states import CashLeadtimeState, CashState, CashStateXR, LeadtimeState, OptDirection, RiskState, State

__all__ = [
    "SdpEngine", "SdpgpuDesc", "SdpgpuError", "SdpgpuStats", "desc_defaults",
    "BackorderFunctor", "LeadtimeFunctor", "CashFunctor", "CashXRFunctor", "OverdraftFunctor", "CashLeadtimeFunctor", "SurvivalFunctor", "CustomFunctor",
    "Recursion", "CLSP", "LeadtimeRecursion", "CashRecursion", "CashRecursionXR", "CashLeadtimeRecursion", "RiskRecursion",
    "StaffRecursion", "StaffFunctor", "StaffState", "BinomialDist", "staff_level_pmf",
    "multilead_solve", "multicash_solve", "multixr_solve", "MultiLeadResult", "Actions", "CashRecursionMulti",
    "CashRecursionMultiLead", "CashRecursionMultiXR", "CashStateMulti", "CashStateMultiLead", "CashStateMultiXR",
    "GetPmf", "PoissonDist", "GammaDist", "NormalDist", "UniformIntDist", "DiscreteDistribution", "Simulation", "RiskSimulation", "Sampling",
    "State", "LeadtimeState", "CashState", "CashStateXR", "CashLeadtimeState", "RiskState", "OptDirection", "java_round",
]
