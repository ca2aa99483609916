// mirror_drivers.cpp -- five driver programs in the shape of the reference's mains, written in C++ over include/sdpgpu_mirror.hpp:
// same local variable names, same lambdas, one constructor swapped (Recursion -> sdp::gpu::Recursion + a
// functor descriptor).  tests/test_gpu_cpp_mirror.py compiles this with g++, runs it on the GPU and
// compares what it prints with the CPU oracle.
//
//   clsp       capacitated.CLSPTesting.main   (src/capacitated/CLSPTesting.java:52-119, one parameter set)
//   leadtime   leadtime.Leadtime.main         (src/leadtime/Leadtime.java:25-99)
//   cash       cash.singleItem.CashConstraint.main (src/cash/singleItem/CashConstraint.java:44-146), smaller grid
//   survival   cash.risk.cashSurvival.main    (src/cash/risk/cashSurvival.java:45-163), smaller grid
//   limit      cash.overdraft.CashOverdraftLimit.main (src/cash/overdraft/CashOverdraftLimit.java:30-113), smaller
//              grid: a driver whose lambdas are NOT a built-in family -- they are passed as HIP device text
//              (argv[3], the file tests/custom_sources.py writes) next to the C++ lambdas
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <iterator>
#include <string>
#include <vector>

#include "sdpgpu_mirror.hpp"

using namespace sdp;

static Pmf read_pmf(const char* path) {
  std::ifstream f(path);
  int T;
  f >> T;
  Pmf pmf((size_t)T);
  for (int t = 0; t < T; ++t) {
    int n;
    f >> n;
    pmf[t].resize((size_t)n);
    for (auto& dp : pmf[t]) f >> dp[0] >> dp[1];
  }
  return pmf;
}

static long java_round(double x) {  // Math.round
  double fl = std::floor(x);
  return (long)((x - fl >= 0.5) ? fl + 1 : fl);
}

static int clsp(const Pmf& pmf) {
  using inventory::State;
  double fixedOrderingCost = 200, variOrderingCost = 1, penaltyCost = 10, holdingCost = 1;
  int maxOrderQuantity = 500;
  double minInventory = -500, maxInventory = 500, stepSize = 1;

  auto getFeasibleAction = [=](const State&) {
    std::vector<double> feasibleActions((size_t)(maxOrderQuantity / stepSize) + 1);
    int index = 0;
    for (double i = 0; i <= maxOrderQuantity; i = i + stepSize) feasibleActions[(size_t)index++] = i;
    return feasibleActions;
  };
  auto stateTransition = [=](const State& state, double action, double randomDemand) {
    double nextInventory = state.getIniInventory() + action - randomDemand;
    nextInventory = nextInventory > maxInventory ? maxInventory : nextInventory;
    nextInventory = nextInventory < minInventory ? minInventory : nextInventory;
    return State(state.getPeriod() + 1, nextInventory);
  };
  auto immediateValue = [=](const State& state, double action, double randomDemand) {
    double fixedCost = action > 0 ? fixedOrderingCost : 0;
    double variableCost = variOrderingCost * action;
    double inventoryLevel = state.getIniInventory() + action - randomDemand;
    double holdingCosts = holdingCost * std::fmax(inventoryLevel, 0);
    double penaltyCosts = penaltyCost * std::fmax(-inventoryLevel, 0);
    return fixedCost + variableCost + holdingCosts + penaltyCosts;
  };

  gpu::BackorderFunctor functor;
  functor.fixedOrderingCost = fixedOrderingCost;
  functor.variOrderingCost = variOrderingCost;
  functor.holdingCost = holdingCost;
  functor.penaltyCost = penaltyCost;
  functor.minInventory = minInventory;
  functor.maxInventory = maxInventory;
  functor.maxOrderQuantity = maxOrderQuantity;
  gpu::Recursion recursion(OptDirection::MIN, pmf, getFeasibleAction, stateTransition, immediateValue, functor);
  int period = 1;
  double iniInventory = 0;
  State initialState(period, iniInventory);
  double finalValue = recursion.getExpectedValue(initialState);
  std::printf("final optimal expected value is: %.17g\n", finalValue);
  std::printf("optimal order quantity in the first priod is : %.17g\n", recursion.getAction(initialState));
  auto optTable = recursion.getOptTable();
  std::printf("optTable rows %zu first %.17g %.17g %.17g\n", optTable.size(), optTable[0][0], optTable[0][1], optTable[0][2]);
  // Simulation.simulateSDPGivenSamplNum's inner loop (Simulation.java:59-69) on one fixed demand path,
  // through the lambdas the recursion hands back
  auto trans = recursion.getStateTransitionFunction();
  auto imm = recursion.getImmediateValueFunction();
  double sum = 0;
  State state = initialState;
  for (size_t t = 0; t < pmf.size(); ++t) {
    double optQ = recursion.getAction(state);
    double randomDemand = (double)java_round(pmf[t][pmf[t].size() / 2][0] + 0.4);
    sum += imm(state, optQ, randomDemand);
    state = trans(state, optQ, randomDemand);
  }
  std::printf("simulated path value %.17g\n", sum);
  return 0;
}

static int leadtime(const Pmf& pmf) {
  using inventory::LeadtimeState;
  double fixedOrderingCost = 0, variOrderingCost = 1, holdingCost = 2, penaltyCost = 10, stepSize = 1;
  int maxOrderQuantity = 100;

  auto getFeasibleAction = [=](const LeadtimeState&) {
    std::vector<double> feasibleActions((size_t)(maxOrderQuantity / stepSize) + 1);
    int index = 0;
    for (double i = 0; i <= maxOrderQuantity; i = i + stepSize) feasibleActions[(size_t)index++] = i;
    return feasibleActions;
  };
  auto stateTransition = [=](const LeadtimeState& s, double action, double randomDemand) {
    double nextInventory = s.getIniInventory() + s.getPreQ() - randomDemand;
    return LeadtimeState(s.getPeriod() + 1, nextInventory, action);
  };
  auto immediateValue = [=](const LeadtimeState& s, double action, double randomDemand) {
    double fixedCost = action > 0 ? fixedOrderingCost : 0;
    double variableCost = variOrderingCost * action;
    double inventoryLevel = s.getIniInventory() + s.getPreQ() - randomDemand;
    double holdingCosts = holdingCost * std::fmax(inventoryLevel, 0);
    double penaltyCosts = penaltyCost * std::fmax(-inventoryLevel, 0);
    return fixedCost + variableCost + holdingCosts + penaltyCosts;
  };
  gpu::LeadtimeFunctor functor;
  functor.fixedOrderingCost = fixedOrderingCost;
  functor.variOrderingCost = variOrderingCost;
  functor.holdingCost = holdingCost;
  functor.penaltyCost = penaltyCost;
  functor.maxOrderQuantity = maxOrderQuantity;
  gpu::LeadtimeRecursion recursion(pmf, getFeasibleAction, stateTransition, immediateValue, functor);
  LeadtimeState initialState(1, 0, 0);
  double opt = recursion.getExpectedValue(initialState);
  std::printf("final optimal expected value is: %.17g\n", opt);
  std::printf("optimal order quantity in the first priod is : %.17g\n", recursion.getAction(initialState));
  auto optTable = recursion.getOptTable();
  std::printf("optTable rows %zu\n", optTable.size());
  return 0;
}

static int cash_constraint(const Pmf& pmf) {
  using cash::CashState;
  const int T = (int)pmf.size();
  double iniInventory = 0, iniCash = 100, fixOrderCost = 0, variCost = 1, price = 10, depositeRate = 0;
  double salvageValue = 0.5 * variCost, holdingCost = 0, overheadCost = 0, overheadRate = 0, maxOrderQuantity = 100;
  int stepSize = 1;
  double minInventoryState = 0, maxInventoryState = 120, minCashState = 0, maxCashState = 600, penaltyCost = 0;
  double discountFactor = 1;

  auto getFeasibleAction = [=](const CashState& s) {
    double maxQ = (int)std::fmin(maxOrderQuantity, std::fmax(0, (s.getIniCash() - overheadCost - fixOrderCost) / variCost));
    std::vector<double> a((size_t)((int)maxQ + 1));
    for (size_t i = 0; i < a.size(); ++i) a[i] = (double)i * stepSize;
    return a;
  };
  auto immediateValue = [=](const CashState& state, double action, double randomDemand) {
    double revenue = price * std::fmin(state.getIniInventory() + action, randomDemand);
    double fixedCost = action > 0 ? fixOrderCost : 0;
    double variableCost = variCost * action;
    double deposite = (state.getIniCash() - fixedCost - variableCost) * (1 + depositeRate);
    double inventoryLevel = state.getIniInventory() + action - randomDemand;
    double holdCosts = holdingCost * std::fmax(inventoryLevel, 0);
    double cashIncrement = (1 - overheadRate) * revenue + deposite - holdCosts - overheadCost - state.getIniCash();
    double salValue = state.getPeriod() == T ? salvageValue * std::fmax(inventoryLevel, 0) : 0;
    cashIncrement += salValue;
    double endCash = state.getIniCash() + cashIncrement;
    if (endCash < 0) cashIncrement += penaltyCost * endCash;
    return cashIncrement;
  };
  auto stateTransition = [=](const CashState& state, double action, double randomDemand) {
    double nextInventory = std::fmax(0, state.getIniInventory() + action - randomDemand);
    double nextCash = state.getIniCash() + immediateValue(state, action, randomDemand);
    nextCash = nextCash > maxCashState ? maxCashState : nextCash;
    nextCash = nextCash < minCashState ? minCashState : nextCash;
    nextInventory = nextInventory > maxInventoryState ? maxInventoryState : nextInventory;
    nextInventory = nextInventory < minInventoryState ? minInventoryState : nextInventory;
    nextCash = java_round(nextCash * 10) / 10.0;
    return CashState(state.getPeriod() + 1, nextInventory, nextCash);
  };
  gpu::CashFunctor functor;
  functor.price = price;
  functor.fixOrderCost = fixOrderCost;
  functor.variCost = variCost;
  functor.holdingCost = holdingCost;
  functor.depositeRate = depositeRate;
  functor.overheadCost = overheadCost;
  functor.overheadRate = overheadRate;
  functor.salvageValue = salvageValue;
  functor.penaltyCost = penaltyCost;
  functor.maxOrderQuantity = maxOrderQuantity;
  functor.minInventoryState = minInventoryState;
  functor.maxInventoryState = maxInventoryState;
  functor.minCashState = minCashState;
  functor.maxCashState = maxCashState;
  functor.iniInventory = iniInventory;
  functor.iniCash = iniCash;
  gpu::CashRecursion recursion(OptDirection::MAX, pmf, getFeasibleAction, stateTransition, immediateValue, discountFactor,
                               functor);
  CashState initialState(1, iniInventory, iniCash);
  recursion.setTreeMapCacheAction();
  double finalValue = recursion.getExpectedValue(initialState);
  std::printf("final optimal cash increment is %.17g\n", finalValue);
  std::printf("optimal order quantity in the first priod is : %.17g\n", recursion.getAction(initialState));
  // one step through the lambdas from the optimal action, then read the table at the successor
  double q = recursion.getAction(initialState);
  CashState next = stateTransition(initialState, q, pmf[0][pmf[0].size() / 2][0]);
  std::printf("successor value %.17g\n", recursion.getExpectedValue(next));
  return 0;
}

static int cash_survival(const Pmf& pmf) {
  using cash::RiskState;
  const int T = (int)pmf.size();
  double iniI = 0, iniCash = 150, fixOrderCost = 0, price = 4, variCost = 1, depositeRate = 0, salvageValue = 0.5;
  double holdingCost = 0, overheadCosts = 100, maxOrderQuantity = 200, stepSize = 1;
  double minInventoryState = 0, maxInventoryState = 200, minCashState = -100, maxCashState = 1500;

  auto getFeasibleAction = [=](const RiskState& s) {
    double maxQ = std::fmin(s.getIniCash() / variCost, maxOrderQuantity);
    if (s.getBankruptBefore() == true) maxQ = 0;
    maxQ = std::fmax(maxQ, 0);
    std::vector<double> a((size_t)((int)maxQ + 1));
    for (size_t i = 0; i < a.size(); ++i) a[i] = (double)i * stepSize;
    return a;
  };
  auto immediateValue = [=](const RiskState& state, double action, double randomDemand) {
    double revenue = price * std::fmin(state.getIniInventory() + action, randomDemand);
    double fixedCost = action > 0 ? fixOrderCost : 0;
    double variableCost = variCost * action;
    double deposite = (state.getIniCash() - fixedCost - variableCost) * (1 + depositeRate);
    double inventoryLevel = state.getIniInventory() + action - randomDemand;
    double holdCosts = holdingCost * std::fmax(inventoryLevel, 0);
    double cashIncrement = revenue + deposite - holdCosts - overheadCosts - state.getIniCash();
    double salValue = state.getPeriod() == T ? salvageValue * std::fmax(inventoryLevel, 0) : 0;
    cashIncrement += salValue;
    return cashIncrement;
  };
  auto stateTransition = [=](const RiskState& state, double action, double randomDemand) {
    double nextInventory = std::fmax(0, state.getIniInventory() + action - randomDemand);
    double nextCash = state.getIniCash() + immediateValue(state, action, randomDemand);
    nextCash = nextCash > maxCashState ? maxCashState : nextCash;
    nextCash = nextCash < minCashState ? minCashState : nextCash;
    nextInventory = nextInventory > maxInventoryState ? maxInventoryState : nextInventory;
    nextInventory = nextInventory < minInventoryState ? minInventoryState : nextInventory;
    nextCash = (double)(java_round(nextCash * 1) / 1);
    return RiskState(state.getPeriod() + 1, nextInventory, nextCash, nextCash < 0);
  };
  gpu::SurvivalFunctor functor;
  functor.price = price;
  functor.fixOrderCost = fixOrderCost;
  functor.variCost = variCost;
  functor.holdingCost = holdingCost;
  functor.depositeRate = depositeRate;
  functor.overheadCost = overheadCosts;
  functor.salvageValue = salvageValue;
  functor.maxOrderQuantity = maxOrderQuantity;
  functor.minInventoryState = minInventoryState;
  functor.maxInventoryState = maxInventoryState;
  functor.minCashState = minCashState;
  functor.maxCashState = maxCashState;
  functor.iniInventory = iniI;
  functor.iniCash = iniCash;
  gpu::RiskRecursion recursion(pmf, getFeasibleAction, stateTransition, immediateValue, functor);
  RiskState initialState(1, iniI, iniCash, false);
  recursion.setTreeMapCacheAction();
  double finalValue = recursion.getSurvProb(initialState);
  std::printf("survival probability for this initial state is: %.17g\n", finalValue);
  std::printf("optimal order quantity in the first priod is : %.17g\n", recursion.getAction(initialState));
  std::printf("visited states %zu\n", recursion.getOptTable().size());
  return 0;
}

static int overdraft_limit(const Pmf& pmf, const char* source_path) {
  using cash::CashState;
  const int T = (int)pmf.size();
  double price = 6, fixOrderCost = 2, variCost = 1, holdingCost = 0.25, interestRate = 0.1, depositeRate = 0;
  double salvageValue = 0.5, maxOrderQuantity = 14, stepSize = 1, discountFactor = 1;
  double minInventoryState = 0, maxInventoryState = 18, minCashState = -40, maxCashState = 120, iniCash = 10;
  std::vector<double> overheadCost = {9, 12, 7, 10};

  auto getFeasibleAction = [=](const CashState&) {
    double maxQ = maxOrderQuantity;
    std::vector<double> a((size_t)((int)maxQ + 1));
    for (size_t i = 0; i < a.size(); ++i) a[i] = (double)i * stepSize;
    return a;
  };
  auto immediateValue = [=](const CashState& state, double action, double randomDemand) {
    double revenue = price * std::fmin(state.getIniInventory() + action, randomDemand);
    double fixedCost = action > 0 ? fixOrderCost : 0;
    double variableCost = variCost * action;
    double inventoryLevel = state.getIniInventory() + action - randomDemand;
    double holdCosts = holdingCost * std::fmax(inventoryLevel, 0);
    double cashBalanceBeforeRevenue =
        state.getIniCash() - fixedCost - variableCost - holdCosts - overheadCost[(size_t)state.getPeriod() - 1];
    double interest = interestRate * std::fmax(-cashBalanceBeforeRevenue, 0);
    double deposite = depositeRate * std::fmax(cashBalanceBeforeRevenue, 0);
    double cashBalanceAfter = cashBalanceBeforeRevenue - interest + deposite + revenue;
    double cashIncrement = cashBalanceAfter - state.getIniCash();
    double salValue = state.getPeriod() == T ? salvageValue * std::fmax(inventoryLevel, 0) : 0;
    cashIncrement += salValue;
    return cashIncrement;
  };
  auto stateTransition = [=](const CashState& state, double action, double randomDemand) {
    double nextInventory = std::fmax(0, state.getIniInventory() + action - randomDemand);
    double nextCash = state.getIniCash() + immediateValue(state, action, randomDemand);
    nextCash = nextCash > maxCashState ? maxCashState : nextCash;
    nextCash = nextCash < minCashState ? minCashState : nextCash;
    nextInventory = nextInventory > maxInventoryState ? maxInventoryState : nextInventory;
    nextInventory = nextInventory < minInventoryState ? minInventoryState : nextInventory;
    nextCash = (double)(java_round(nextCash * 10) / 10);
    return CashState(state.getPeriod() + 1, nextInventory, nextCash);
  };
  gpu::CashFunctor functor;  // state shape and grid only: the formulas come from the device text
  functor.overdraft = true;
  functor.maxOrderQuantity = maxOrderQuantity;
  functor.minInventoryState = minInventoryState;
  functor.maxInventoryState = maxInventoryState;
  functor.minCashState = minCashState;
  functor.maxCashState = maxCashState;
  functor.cashRoundIntDiv = true;  // Math.round(nextCash * 10) / 10
  functor.iniCash = iniCash;
  std::ifstream f(source_path);
  functor.user.source.assign(std::istreambuf_iterator<char>(f), std::istreambuf_iterator<char>());
  functor.user.params = {price, fixOrderCost, variCost, holdingCost, interestRate, depositeRate, salvageValue,
                         maxOrderQuantity, minInventoryState, maxInventoryState, minCashState, maxCashState};
  for (int t = 0; t < T; ++t) functor.user.params.push_back(overheadCost[(size_t)t]);
  gpu::CashRecursion recursion(OptDirection::MAX, pmf, getFeasibleAction, stateTransition, immediateValue, discountFactor,
                               functor);
  CashState initialState(1, 0, iniCash);
  recursion.setTreeMapCacheAction();
  double finalCash = iniCash + recursion.getExpectedValue(initialState);
  std::printf("final optimal cash is: %.17g\n", finalCash);
  std::printf("optimal order quantity in the first priod is : %.17g\n", recursion.getAction(initialState));
  double q = recursion.getAction(initialState);
  CashState next = stateTransition(initialState, q, pmf[0][pmf[0].size() / 2][0]);
  std::printf("successor value %.17g\n", recursion.getExpectedValue(next));
  return 0;
}

// workforce.WorkforcePlanning.main (src/workforce/WorkforcePlanning.java:32-118) on a smaller staff range; the
// binomial table comes from the file (rows of i + 1 probabilities per period), SSJ is not available here.
static int workforce_planning(const char* path) {
  using workforce::StaffState;
  std::ifstream in(path);
  int T, xLength;
  in >> T >> xLength;
  LevelPmf pmf((size_t)T);
  for (int t = 0; t < T; t++) {
    pmf[t].resize((size_t)xLength);
    for (int i = 0; i < xLength; i++) {
      pmf[t][i].resize((size_t)i + 1);
      for (int j = 0; j < i + 1; j++) {
        pmf[t][i][j][0] = j;
        in >> pmf[t][i][j][1];
      }
    }
  }
  int iniStaffNum = 0;
  double fixCost = 100, unitVariCost = 10, salary = 20, unitPenalty = 80;
  std::vector<int> minStaffNum((size_t)T, 12);
  int maxHireNum = 40, stepSize = 1, minX = 0, maxX = xLength - 1;

  auto getFeasibleAction = [=](const StaffState&) {
    std::vector<int> feasibleActions((size_t)(maxHireNum / stepSize) + 1);
    int index = 0;
    for (int i = 0; i <= maxHireNum; i = i + stepSize) feasibleActions[(size_t)index++] = i;
    return feasibleActions;
  };
  auto stateTransition = [=](const StaffState& state, int action, int randomDemand) {
    int nextStaffNum = state.iniStaffNum + action - randomDemand;
    nextStaffNum = nextStaffNum > maxX ? maxX : nextStaffNum;
    nextStaffNum = nextStaffNum < minX ? minX : nextStaffNum;
    return StaffState(state.period + 1, nextStaffNum);
  };
  auto immediateValue = [=](const StaffState& state, int action, int randomDemand) {
    double fixHireCost = action > 0 ? fixCost : 0;
    double variHireCost = unitVariCost * action;
    int nextStaffNum = state.iniStaffNum + action - randomDemand;
    double salaryCost = salary * nextStaffNum;
    int t = state.period - 1;
    double penaltyCost = nextStaffNum > minStaffNum[(size_t)t] ? 0 : unitPenalty * (minStaffNum[(size_t)t] - nextStaffNum);
    double totalCosts = fixHireCost + variHireCost + salaryCost + penaltyCost;
    return totalCosts;
  };

  gpu::StaffFunctor functor;
  functor.fixCost = fixCost;
  functor.unitVariCost = unitVariCost;
  functor.salary = salary;
  functor.unitPenalty = unitPenalty;
  functor.minStaffNum = minStaffNum;
  functor.maxHireNum = maxHireNum;
  functor.minX = minX;
  functor.maxX = maxX;
  functor.iniStaffNum = iniStaffNum;
  gpu::StaffRecursion recursion(getFeasibleAction, stateTransition, immediateValue, pmf, T, functor);
  int period = 1;
  StaffState initialState(period, iniStaffNum);
  double opt = recursion.getExpectedValue(initialState);
  std::printf("final optimal expected cost is: %.17g\n", opt);
  int optQ = recursion.getAction(initialState);
  std::printf("optimal hiring number in the first priod is : %d\n", optQ);
  auto optTable = recursion.getOptTable();
  std::printf("visited states: %zu\n", optTable.size());
  // the lambdas handed in are the ones the device family restates: spot-check one cell on the host
  StaffState s2 = recursion.getStateTransitionFunction()(initialState, optQ, 3);
  std::printf("V_2 after the first decision and 3 leavers: %.17g\n", recursion.getExpectedValue(s2));
  return 0;
}

// cash.multiItem.MultiItemCashXR.main (src/cash/multiItem/MultiItemCashXR.java:41-164) on a smaller box: the joint
// pmf comes from the file (T, then per period n and n rows {d1, d2, p}); SSJ is not available here.
static int multi_item_cash_xr(const char* path) {
  std::ifstream in(path);
  int T;
  in >> T;
  gpu::MultiPmf pmf((size_t)T);
  for (int t = 0; t < T; t++) {
    int n;
    in >> n;
    pmf[t].resize((size_t)n);
    for (auto& r : pmf[t]) in >> r[0] >> r[1] >> r[2];
  }
  double iniCash = 0;
  int iniInventory1 = 0, iniInventory2 = 0;
  gpu::MultiItemFunctor functor;
  functor.Qbound = 20;
  functor.price = {5, 10};
  functor.variCost = {1, 2};
  functor.salPrice = {0.5, 1.0};
  functor.minInventoryState = 0;
  functor.maxInventoryState = 200;
  functor.minCashState = 0;
  functor.maxCashState = 10000;
  functor.depositeRate = 0;
  double discountFactor = 1;
  gpu::CashRecursionMultiXR recursion(discountFactor, pmf, T, functor);
  int period = 1;
  gpu::CashRecursionMultiXR::State iniState{period, (double)iniInventory1, (double)iniInventory2, iniCash};
  double finalValue = iniCash + recursion.getExpectedValue(iniState);
  std::printf("final optimal cash  is %.17g\n", finalValue);
  auto y = recursion.getAction(iniState);
  std::printf("optimal order quantity in the first priod is :  y1 = %d, y2 = %d\n", y[0], y[1]);
  std::printf("visited states: %zu\n", recursion.getCacheActions().size());
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 3) return 1;
  try {
    const std::string which = argv[1];
    if (which == "workforce") return workforce_planning(argv[2]);
    if (which == "multixr") return multi_item_cash_xr(argv[2]);
    const Pmf pmf = read_pmf(argv[2]);
    if (which == "clsp") return clsp(pmf);
    if (which == "leadtime") return leadtime(pmf);
    if (which == "cash") return cash_constraint(pmf);
    if (which == "survival") return cash_survival(pmf);
    if (which == "limit" && argc > 3) return overdraft_limit(pmf, argv[3]);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 2;
  }
  return 1;
}
