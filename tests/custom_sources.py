"""Lambdas of reference drivers written as HIP device text for sdpgpu_create_custom.  The same text is compiled
for the host by the oracle harness (oracle/sdpref.py: custom_functor), so both sides run the same three functions
and the test compares the ENGINES around them (loop, layout, arg-opt, discount)."""

# capacitated.CLSP's lambdas (CLSP.java:251-272); params = {K, v, h, pi, minInventory, maxInventory, maxOrderQuantity}
BACKORDER = r"""
__device__ int sdp_feasible_count(const sdp_ctx& c, double x, double cash, double preq) {
  return (int)(c.params[6] / c.step) + 1;
}
__device__ double sdp_immediate(const sdp_ctx& c, double x, double cash, double preq, double action, double randomDemand) {
  double fixedCost = action > 0 ? c.params[0] : 0;
  double variableCost = c.params[1] * action;
  double inventoryLevel = x + action - randomDemand;
  double holdingCosts = c.params[2] * sdp_max(inventoryLevel, 0);
  double penaltyCosts = c.params[3] * sdp_max(-inventoryLevel, 0);
  double totalCosts = fixedCost + variableCost + holdingCosts + penaltyCosts;
  return totalCosts;
}
__device__ void sdp_transition(const sdp_ctx& c, double x, double cash, double preq, double action, double randomDemand,
                               double& nx, double& ncash, double& npreq) {
  double nextInventory = x + action - randomDemand;
  nextInventory = nextInventory > c.params[5] ? c.params[5] : nextInventory;
  nextInventory = nextInventory < c.params[4] ? c.params[4] : nextInventory;
  nx = nextInventory;
  ncash = 0;
  npreq = 0;
}
"""

# leadtime.Leadtime's lambdas (Leadtime.java:50-81, inventory clamp added); params = {K, v, h, pi, min, max, maxQ}
LEADTIME = r"""
__device__ int sdp_feasible_count(const sdp_ctx& c, double x, double cash, double preq) {
  return (int)(c.params[6] / c.step) + 1;
}
__device__ double sdp_immediate(const sdp_ctx& c, double x, double cash, double preq, double action, double randomDemand) {
  double fixedCost = action > 0 ? c.params[0] : 0;
  double variableCost = c.params[1] * action;
  double inventoryLevel = x + preq - randomDemand;
  double holdingCosts = c.params[2] * sdp_max(inventoryLevel, 0);
  double penaltyCosts = c.params[3] * sdp_max(-inventoryLevel, 0);
  return fixedCost + variableCost + holdingCosts + penaltyCosts;
}
__device__ void sdp_transition(const sdp_ctx& c, double x, double cash, double preq, double action, double randomDemand,
                               double& nx, double& ncash, double& npreq) {
  double nextInventory = x + preq - randomDemand;
  nextInventory = nextInventory > c.params[5] ? c.params[5] : nextInventory;
  nextInventory = nextInventory < c.params[4] ? c.params[4] : nextInventory;
  nx = nextInventory;
  ncash = 0;
  npreq = action;
}
"""

# cash.overdraft.CashOverdraftLimit's lambdas (CashOverdraftLimit.java:61-99) -- NOT one of the built-in families:
# simple interest / deposit on the balance before revenue, holding cost inside that balance, per-period overhead,
# `Math.round(nextCash * 10) / 10` (long division).
# params = {price, fixOrderCost, variCost, holdingCost, interestRate, depositeRate, salvageValue, maxOrderQuantity,
#           minInventoryState, maxInventoryState, minCashState, maxCashState, overheadCost[0..T-1]}
OVERDRAFT_LIMIT = r"""
__device__ int sdp_feasible_count(const sdp_ctx& c, double x, double cash, double preq) {
  double maxQ = c.params[7];  // CashOverdraftLimit.java:65: `maxQ = maxOrderQuantity` overrides the cash bound
  return (int)maxQ + 1;
}
__device__ double sdp_immediate(const sdp_ctx& c, double x, double cash, double preq, double action, double randomDemand) {
  const double price = c.params[0], fixOrderCost = c.params[1], variCost = c.params[2], holdingCost = c.params[3];
  const double interestRate = c.params[4], depositeRate = c.params[5], salvageValue = c.params[6];
  double revenue = price * sdp_min(x + action, randomDemand);
  double fixedCost = action > 0 ? fixOrderCost : 0;
  double variableCost = variCost * action;
  double inventoryLevel = x + action - randomDemand;
  double holdCosts = holdingCost * sdp_max(inventoryLevel, 0);
  double cashBalanceBeforeRevenue = cash - fixedCost - variableCost - holdCosts - c.params[12 + c.period - 1];
  double interest = interestRate * sdp_max(-cashBalanceBeforeRevenue, 0);
  double deposite = depositeRate * sdp_max(cashBalanceBeforeRevenue, 0);
  double cashBalanceAfter = cashBalanceBeforeRevenue - interest + deposite + revenue;
  double cashIncrement = cashBalanceAfter - cash;
  double salValue = c.period == c.T ? salvageValue * sdp_max(inventoryLevel, 0) : 0;
  cashIncrement += salValue;
  return cashIncrement;
}
__device__ void sdp_transition(const sdp_ctx& c, double x, double cash, double preq, double action, double randomDemand,
                               double& nx, double& ncash, double& npreq) {
  double nextInventory = sdp_max(0, x + action - randomDemand);
  double nextCash = cash + sdp_immediate(c, x, cash, preq, action, randomDemand);
  nextCash = nextCash > c.params[11] ? c.params[11] : nextCash;
  nextCash = nextCash < c.params[10] ? c.params[10] : nextCash;
  nextInventory = nextInventory > c.params[9] ? c.params[9] : nextInventory;
  nextInventory = nextInventory < c.params[8] ? c.params[8] : nextInventory;
  nextCash = sdp_trunc(sdp_round(nextCash * 10) / 10);  // Math.round(nextCash * 10) / 10: long / int
  nx = nextInventory;
  ncash = nextCash;
  npreq = 0;
}
"""

# the same lambdas with the FUSED callback (ABI 5): the increment is formed once per cell and both the immediate value and the
# successor are read off it; sdp_immediate / sdp_transition are calls of sdp_cell
OVERDRAFT_LIMIT_FUSED = r"""
#define SDP_USER_CELL 1
__device__ int sdp_feasible_count(const sdp_ctx& c, double x, double cash, double preq) {
  double maxQ = c.params[7];
  return (int)maxQ + 1;
}
__device__ void sdp_cell(const sdp_ctx& c, double x, double cash, double preq, double action, double randomDemand,
                         double& imm, double& nx, double& ncash, double& npreq) {
  const double price = c.params[0], fixOrderCost = c.params[1], variCost = c.params[2], holdingCost = c.params[3];
  const double interestRate = c.params[4], depositeRate = c.params[5], salvageValue = c.params[6];
  double revenue = price * sdp_min(x + action, randomDemand);
  double fixedCost = action > 0 ? fixOrderCost : 0;
  double variableCost = variCost * action;
  double inventoryLevel = x + action - randomDemand;
  double holdCosts = holdingCost * sdp_max(inventoryLevel, 0);
  double cashBalanceBeforeRevenue = cash - fixedCost - variableCost - holdCosts - c.params[12 + c.period - 1];
  double interest = interestRate * sdp_max(-cashBalanceBeforeRevenue, 0);
  double deposite = depositeRate * sdp_max(cashBalanceBeforeRevenue, 0);
  double cashBalanceAfter = cashBalanceBeforeRevenue - interest + deposite + revenue;
  double cashIncrement = cashBalanceAfter - cash;
  double salValue = c.period == c.T ? salvageValue * sdp_max(inventoryLevel, 0) : 0;
  cashIncrement += salValue;
  imm = cashIncrement;
  double nextInventory = sdp_max(0, inventoryLevel);
  double nextCash = cash + cashIncrement;
  nextCash = nextCash > c.params[11] ? c.params[11] : nextCash;
  nextCash = nextCash < c.params[10] ? c.params[10] : nextCash;
  nextInventory = nextInventory > c.params[9] ? c.params[9] : nextInventory;
  nextInventory = nextInventory < c.params[8] ? c.params[8] : nextInventory;
  nextCash = sdp_trunc(sdp_round(nextCash * 10) / 10);
  nx = nextInventory;
  ncash = nextCash;
  npreq = 0;
}
__device__ double sdp_immediate(const sdp_ctx& c, double x, double cash, double preq, double action, double randomDemand) {
  double imm, nx, nc, nq;
  sdp_cell(c, x, cash, preq, action, randomDemand, imm, nx, nc, nq);
  return imm;
}
__device__ void sdp_transition(const sdp_ctx& c, double x, double cash, double preq, double action, double randomDemand,
                               double& nx, double& ncash, double& npreq) {
  double imm;
  sdp_cell(c, x, cash, preq, action, randomDemand, imm, nx, ncash, npreq);
}
"""

# a transition that forgets to clamp: leaves the grid
BROKEN_TRANSITION = BACKORDER.replace("nextInventory = nextInventory < c.params[4] ? c.params[4] : nextInventory;", "")


# the same two texts with the long division `Math.round(nextCash * 10) / 10` written through the prelude's integer helper
# (sdp_ldiv: truncating integer division, a few integer instructions with a literal divisor) instead of an fp64 division
OVERDRAFT_LIMIT_LDIV = OVERDRAFT_LIMIT.replace("sdp_trunc(sdp_round(nextCash * 10) / 10)", "sdp_ldiv(sdp_round(nextCash * 10), 10)")
OVERDRAFT_LIMIT_FUSED_LDIV = OVERDRAFT_LIMIT_FUSED.replace("sdp_trunc(sdp_round(nextCash * 10) / 10)", "sdp_ldiv(sdp_round(nextCash * 10), 10)")
assert "sdp_ldiv" in OVERDRAFT_LIMIT_LDIV and "sdp_ldiv" in OVERDRAFT_LIMIT_FUSED_LDIV
