// sdpgpu_mirror.hpp -- C++17 host-side mirror of the reference's recursion classes over the C ABI of
// sdpgpu.h (header-only; link with -lsdpgpu).
//
// The reference is Java and the build image has no JDK, so this is the compiled-language host side:
// the same class names, constructor argument order, method names and argument meaning as
//
//     sdp.inventory.State / LeadtimeState          src/sdp/inventory/State.java:12-76, LeadtimeState.java:10-52
//     sdp.cash.CashState / CashLeadtimeState       src/sdp/cash/CashState.java:12-48, CashLeadtimeState.java:11-46
//     sdp.inventory.Recursion                      src/sdp/inventory/Recursion.java:33-188
//     sdp.inventory.LeadtimeRecursion              src/sdp/inventory/LeadtimeRecursion.java:19-104
//     sdp.cash.CashRecursion                       src/sdp/cash/CashRecursion.java:23-218
//     sdp.cash.CashLeadtimeRecursion               src/sdp/cash/CashLeadtimeRecursion.java:19-107
//     StateTransitionFunction / ImmediateValueFunction   StateTransition.java:20-22, ImmediateValue.java:23-25
//
// plus ONE extra constructor argument: the functor descriptor naming the closed-form family the three
// lambdas belong to (a GPU cannot call host closures per cell).  The lambdas are kept and handed back
// by getStateTransitionFunction / getImmediateValueFunction exactly as the simulators expect
// (Simulation.java:39-40).  getExpectedValue(state) runs the whole backward sweep on the GPU on first
// use.  Errors of the C ABI become std::runtime_error carrying sdpgpu_last_error().
#pragma once

#include <algorithm>
#include <array>
#include <cmath>
#include <functional>
#include <map>
#include <stdexcept>
#include <string>
#include <tuple>
#include <vector>

#include "sdpgpu.h"

namespace sdp {

enum class OptDirection { MIN, MAX };  // Recursion.java:44-47, CashRecursion.java:34-37

/** double[][][] pmf: pmf[t][j] = {demand, probability} (Recursion.java:38). */
using Pmf = std::vector<std::vector<std::array<double, 2>>>;

/** double[][][][] pmfs of workforce.StaffRecursion: pmfs[t][y][j] = {j, P(turnover j | hire-up-to level y)}
 *  (StaffRecursion.java:23, WorkforcePlanning.java:52-69). */
using LevelPmf = std::vector<std::vector<std::vector<std::array<double, 2>>>>;

template <class S, class A, class R, class S2>
using StateTransitionFunction = std::function<S2(const S&, A, R)>;
template <class S, class A, class R, class V>
using ImmediateValueFunction = std::function<V(const S&, A, R)>;

namespace inventory {

class State {
 public:
  State(int period, double initialInventory) : period(period), initialInventory(initialInventory) {}
  int getPeriod() const { return period; }
  double getIniInventory() const { return initialInventory; }
  bool operator==(const State& o) const { return period == o.period && initialInventory == o.initialInventory; }

 protected:
  int period;
  double initialInventory;
};

class LeadtimeState : public State {
 public:
  LeadtimeState(int period, double initialInventory, double preQ) : State(period, initialInventory), preQ(preQ) {}
  double getPreQ() const { return preQ; }
  bool operator==(const LeadtimeState& o) const { return State::operator==(o) && preQ == o.preQ; }

 private:
  double preQ;
};

}  // namespace inventory

namespace cash {

class CashState : public inventory::State {
 public:
  CashState(int period, double initialInventory, double iniCash) : State(period, initialInventory), iniCash(iniCash) {}
  double getIniCash() const { return iniCash; }
  bool operator==(const CashState& o) const { return State::operator==(o) && iniCash == o.iniCash; }
  double iniCash;
};

class RiskState : public CashState {  // RiskState.java:12-52 (the constructor there stores `false`, :17)
 public:
  RiskState(int period, double initialInventory, double iniCash, bool /*bankruptBefore*/)
      : CashState(period, initialInventory, iniCash) {}
  bool getBankruptBefore() const { return false; }
};

class CashLeadtimeState : public CashState {
 public:
  CashLeadtimeState(int period, double initialInventory, double iniCash, double preQ)
      : CashState(period, initialInventory, iniCash), preQ(preQ) {}
  double getPreQ() const { return preQ; }

 private:
  double preQ;
};

}  // namespace cash

namespace workforce {

class StaffState {  // StaffState.java:4-35
 public:
  StaffState(int period, int iniStaffNum) : period(period), iniStaffNum(iniStaffNum) {}
  bool operator==(const StaffState& o) const { return period == o.period && iniStaffNum == o.iniStaffNum; }
  int period, iniStaffNum;
};

}  // namespace workforce

namespace gpu {

// ---- functor descriptors: field names are the reference's local variable names -----------------------
// The driver's own lambdas as HIP device text (sdpgpu_create_custom, include/sdpgpu.h): when `source` is set the
// engine compiles it with hipRTC and runs it instead of the functor's built-in family; the functor then only
// describes the state shape and the grid.  `params` = the constants the Java lambdas close over (c.params[]).
struct UserLambdas {
  std::string source;
  std::vector<double> params;
};

struct BackorderFunctor {  // CLSP.java:251-272, CLSPTesting.java:78-106
  UserLambdas user;
  double fixedOrderingCost = 0, variOrderingCost = 0, holdingCost = 0, penaltyCost = 0;
  double minInventory = 0, maxInventory = 0, maxOrderQuantity = 0, stepSize = 1, iniInventory = 0;
  void fill(sdpgpu_desc& d) const {
    d.family = SDPGPU_FAMILY_BACKORDER;
    d.step = stepSize;
    d.min_inventory = minInventory;
    d.max_inventory = maxInventory;
    d.max_order_quantity = maxOrderQuantity;
    d.fixed_order_cost = fixedOrderingCost;
    d.unit_order_cost = variOrderingCost;
    d.holding_cost = holdingCost;
    d.penalty_cost = penaltyCost;
    d.ini_inventory = iniInventory;
  }
};

struct LeadtimeFunctor {  // Leadtime.java:50-81
  UserLambdas user;
  double fixedOrderingCost = 0, variOrderingCost = 0, holdingCost = 0, penaltyCost = 0;
  double maxOrderQuantity = 0, stepSize = 1, iniInventory = 0, iniPreQ = 0;
  bool clampInventory = false;  // Leadtime.java:65-66 has the clamp commented out
  double minInventory = 0, maxInventory = 0;
  void fill(sdpgpu_desc& d) const {
    d.family = SDPGPU_FAMILY_LEADTIME;
    d.step = stepSize;
    d.clamp_inventory = clampInventory ? 1 : 0;
    d.min_inventory = minInventory;
    d.max_inventory = maxInventory;
    d.max_order_quantity = maxOrderQuantity;
    d.fixed_order_cost = fixedOrderingCost;
    d.unit_order_cost = variOrderingCost;
    d.holding_cost = holdingCost;
    d.penalty_cost = penaltyCost;
    d.ini_inventory = iniInventory;
    d.ini_preq = iniPreQ;
  }
};

struct CashFunctor {  // CashConstraint.java:95-133 (cashFormula 0), CashConstraintTesting.java:110-148 (1)
  UserLambdas user;
  double price = 0, fixOrderCost = 0, variCost = 1, holdingCost = 0, depositeRate = 0, overheadCost = 0;
  double overheadRate = 0, salvageValue = 0, penaltyCost = 0, maxOrderQuantity = 0, stepSize = 1;
  double minInventoryState = 0, maxInventoryState = 0, minCashState = 0, maxCashState = 0;
  double cashRoundMult = 10, cashRoundDiv = 10;  // Math.round(nextCash * 10) / 10.0
  bool cashRoundIntDiv = false;                  // `/ 10` (long division), CashOverdraft.java:116
  int cashFormula = 0;
  double iniInventory = 0, iniCash = 0;
  // overdraft schedule (CashOverdraft.java:86-95); family switches to OVERDRAFT when `overdraft` is set
  bool overdraft = false;
  double r0 = 0, r2 = 0, r3 = 0, limit = 0, interestFreeAmount = 0;
  std::vector<double> overheadCosts;  // per period, optional
  void fill(sdpgpu_desc& d) const {
    d.family = overdraft ? SDPGPU_FAMILY_OVERDRAFT : SDPGPU_FAMILY_CASH;
    d.step = stepSize;
    d.min_inventory = minInventoryState;
    d.max_inventory = maxInventoryState;
    d.max_order_quantity = maxOrderQuantity;
    d.fixed_order_cost = fixOrderCost;
    d.unit_order_cost = variCost;
    d.holding_cost = holdingCost;
    d.penalty_cost = penaltyCost;
    d.price = price;
    d.salvage_value = salvageValue;
    d.deposit_rate = depositeRate;
    d.overhead_cost = overheadCost;
    d.overhead_rate = overheadRate;
    d.min_cash = minCashState;
    d.max_cash = maxCashState;
    d.cash_round_mult = cashRoundMult;
    d.cash_round_div = cashRoundDiv;
    d.cash_round_int_div = cashRoundIntDiv ? 1 : 0;
    d.cash_formula = cashFormula;
    d.ini_inventory = iniInventory;
    d.ini_cash = iniCash;
    d.r0 = r0;
    d.r2 = r2;
    d.r3 = r3;
    d.overdraft_limit = limit;
    d.interest_free_amount = interestFreeAmount;
  }
};

struct SurvivalFunctor : CashFunctor {  // cashSurvival.java:98-143 under RiskRecursion.getSurvProb
  SurvivalFunctor() {
    cashRoundMult = 1;  // Math.round(nextCash * 1) / 1
    cashRoundDiv = 1;
    cashRoundIntDiv = true;
  }
  void fill(sdpgpu_desc& d) const {
    CashFunctor::fill(d);
    d.family = SDPGPU_FAMILY_SURVIVAL;
  }
};

struct CashLeadtimeFunctor : CashFunctor {  // SingleProductLeadtime.java:72-119
  double iniPreQ = 0;
  bool zeroOrderLastPeriod = true;
  CashLeadtimeFunctor() {
    cashRoundMult = 100;
    cashRoundDiv = 100;
  }
  void fill(sdpgpu_desc& d) const {
    CashFunctor::fill(d);
    d.family = SDPGPU_FAMILY_CASH_LEADTIME;
    d.ini_preq = iniPreQ;
    d.zero_order_last_period = zeroOrderLastPeriod ? 1 : 0;
  }
};

// ---- RAII wrapper of one sdpgpu_handle ---------------------------------------------------------------
class Engine {
 public:
  Engine(sdpgpu_desc desc, const Pmf& pmf, const std::vector<double>& overhead = {}, const UserLambdas& user = {})
      : T_((int)pmf.size()) {
    desc.periods = T_;
    const int rc = user.source.empty()
                       ? sdpgpu_create(&desc, &h_)
                       : sdpgpu_create_custom(&desc, user.source.c_str(), user.params.data(), (int32_t)user.params.size(), &h_);
    if (rc != 0) throw std::runtime_error(sdpgpu_last_error(nullptr));
    try {
      for (int t = 0; t < T_; ++t) {
        std::vector<double> d, p;
        for (const auto& dp : pmf[t]) {
          d.push_back(dp[0]);
          p.push_back(dp[1]);
        }
        check(sdpgpu_set_pmf(h_, t, d.data(), p.data(), (int32_t)d.size()));
      }
      for (size_t t = 0; t < overhead.size(); ++t) check(sdpgpu_set_overhead(h_, (int32_t)t, overhead[t]));
    } catch (...) {
      sdpgpu_destroy(h_);
      throw;
    }
  }
  /** STAFF family: the level-dependent table instead of one pmf per period; minStaffNum[t] rides on set_overhead. */
  Engine(sdpgpu_desc desc, const LevelPmf& pmfs, const std::vector<int>& minStaffNum) : T_((int)pmfs.size()) {
    desc.periods = T_;
    if (sdpgpu_create(&desc, &h_) != 0) throw std::runtime_error(sdpgpu_last_error(nullptr));
    try {
      for (int t = 0; t < T_; ++t) {
        const int32_t rows = (int32_t)pmfs[t].size();
        int32_t stride = 1;
        std::vector<int32_t> len;
        for (const auto& row : pmfs[t]) {
          len.push_back((int32_t)row.size());
          stride = std::max(stride, (int32_t)row.size());
        }
        std::vector<double> prob((size_t)rows * (size_t)stride, 0.0);
        for (int32_t y = 0; y < rows; ++y)
          for (size_t j = 0; j < pmfs[t][y].size(); ++j) {
            if (pmfs[t][y][j][0] != (double)j) throw std::invalid_argument("pmfs[t][y][j][0] must be j");
            prob[(size_t)y * (size_t)stride + j] = pmfs[t][y][j][1];
          }
        check(sdpgpu_set_level_pmf(h_, t, prob.data(), len.data(), rows, stride));
        check(sdpgpu_set_overhead(h_, t, (double)minStaffNum.at((size_t)t)));
      }
    } catch (...) {
      sdpgpu_destroy(h_);
      throw;
    }
  }
  ~Engine() { sdpgpu_destroy(h_); }
  Engine(const Engine&) = delete;
  Engine& operator=(const Engine&) = delete;

  int periods() const { return T_; }
  sdpgpu_handle* handle() { return h_; }
  void check(int rc) const {
    if (rc != 0) throw std::runtime_error(sdpgpu_last_error(h_));
  }
  void solve() {
    if (!solved_) {
      check(sdpgpu_solve(h_, 1));
      solved_ = true;
    }
  }
  /** (value, action index) of a state: from the dense tables, or evaluated on the device if off-grid. */
  std::pair<double, int> lookup(int period, double x, double cash, double preq) {
    solve();
    if (period < 1 || period > T_) throw std::out_of_range("period");
    const int64_t idx = sdpgpu_state_index(h_, period, x, cash, preq);
    if (idx >= 0) {
      table(period);
      return {values_[period - 1][(size_t)idx], policy_[period - 1][(size_t)idx]};
    }
    double v;
    int32_t a;
    check(sdpgpu_eval_states(h_, period, 1, &x, &cash, &preq, &v, &a));
    return {v, a};
  }
  void table(int period) {
    if (values_.empty()) {
      values_.resize((size_t)T_);
      policy_.resize((size_t)T_);
    }
    if (!values_[period - 1].empty()) return;
    const int64_t n = sdpgpu_num_states(h_, period);
    values_[period - 1].resize((size_t)n);
    policy_[period - 1].resize((size_t)n);
    check(sdpgpu_values(h_, period, values_[period - 1].data(), n));
    check(sdpgpu_policy(h_, period, policy_[period - 1].data(), 0, n));
  }
  const std::vector<int32_t>& policy(int period) {
    solve();
    table(period);
    return policy_[period - 1];
  }
  std::vector<uint8_t> reachable(int period) {
    solve();
    std::vector<uint8_t> m((size_t)sdpgpu_num_states(h_, period));
    check(sdpgpu_reachable(h_, period, m.data(), (int64_t)m.size()));
    return m;
  }

 private:
  sdpgpu_handle* h_ = nullptr;
  int T_;
  bool solved_ = false;
  std::vector<std::vector<double>> values_;
  std::vector<std::vector<int32_t>> policy_;
};

inline sdpgpu_desc make_desc(OptDirection dir) {
  sdpgpu_desc d;
  sdpgpu_desc_init(&d);
  d.direction = dir == OptDirection::MIN ? SDPGPU_MIN : SDPGPU_MAX;
  return d;
}

// ---- sdp.inventory.Recursion -------------------------------------------------------------------------
class Recursion {
 public:
  using State = inventory::State;
  using Trans = StateTransitionFunction<State, double, double, State>;
  using Imm = ImmediateValueFunction<State, double, double, double>;
  using Actions = std::function<std::vector<double>(const State&)>;

  Recursion(OptDirection optDirection, const Pmf& pmf, Actions getFeasibleAction, Trans stateTransition,
            Imm immediateValue, const BackorderFunctor& functor)
      : getFeasibleActions(std::move(getFeasibleAction)), stateTransition(std::move(stateTransition)),
        immediateValue(std::move(immediateValue)), step_(functor.stepSize), engine_(desc_of(optDirection, functor), pmf, {}, functor.user) {}

  Trans getStateTransitionFunction() const { return stateTransition; }
  Imm getImmediateValueFunction() const { return immediateValue; }
  void setTreeMapCacheAction() {}  // the dense tables are already in comparator order (Recursion.java:80-86)

  double getExpectedValue(const State& s) { return engine_.lookup(s.getPeriod(), s.getIniInventory(), 0, 0).first; }
  double getAction(const State& s) { return engine_.lookup(s.getPeriod(), s.getIniInventory(), 0, 0).second * step_; }

  /** rows {period, inventory, Q} of the reachable states in (period, inventory) order (Recursion.java:177-186). */
  std::vector<std::array<double, 3>> getOptTable() {
    std::vector<std::array<double, 3>> rows;
    for (int period = 1; period <= engine_.periods(); ++period) {
      const auto mask = engine_.reachable(period);
      const auto& pol = engine_.policy(period);
      double x_lo;
      int64_t nx, nc, nq;
      engine_.check(sdpgpu_grid(engine_.handle(), period, &x_lo, &nx, &nc, &nq));
      for (size_t i = 0; i < mask.size(); ++i)
        if (mask[i]) rows.push_back({(double)period, x_lo + (double)i * step_, pol[i] * step_});
    }
    return rows;
  }
  Engine& engine() { return engine_; }

  Actions getFeasibleActions;
  Trans stateTransition;
  Imm immediateValue;

 private:
  static sdpgpu_desc desc_of(OptDirection dir, const BackorderFunctor& f) {
    sdpgpu_desc d = make_desc(dir);
    f.fill(d);
    return d;
  }
  double step_;
  Engine engine_;
};

// ---- sdp.inventory.LeadtimeRecursion (MIN only, LeadtimeRecursion.java:52,66) ------------------------
class LeadtimeRecursion {
 public:
  using State = inventory::LeadtimeState;
  using Trans = StateTransitionFunction<State, double, double, State>;
  using Imm = ImmediateValueFunction<State, double, double, double>;
  using Actions = std::function<std::vector<double>(const State&)>;

  LeadtimeRecursion(const Pmf& pmf, Actions getFeasibleAction, Trans stateTransition, Imm immediateValue,
                    const LeadtimeFunctor& functor)
      : getFeasibleActions(std::move(getFeasibleAction)), stateTransition(std::move(stateTransition)),
        immediateValue(std::move(immediateValue)), step_(functor.stepSize), engine_(desc_of(functor), pmf, {}, functor.user) {}

  double getExpectedValue(const State& s) {
    return engine_.lookup(s.getPeriod(), s.getIniInventory(), 0, s.getPreQ()).first;
  }
  double getAction(const State& s) {
    return engine_.lookup(s.getPeriod(), s.getIniInventory(), 0, s.getPreQ()).second * step_;
  }
  /** rows {period, inventory, preQ, Q} ordered by (period, inventory, preQ) (LeadtimeRecursion.java:37-40,93-102). */
  std::vector<std::array<double, 4>> getOptTable() {
    std::vector<std::array<double, 4>> rows;
    for (int period = 1; period <= engine_.periods(); ++period) {
      const auto mask = engine_.reachable(period);
      const auto& pol = engine_.policy(period);
      double x_lo;
      int64_t nx, nc, nq;
      engine_.check(sdpgpu_grid(engine_.handle(), period, &x_lo, &nx, &nc, &nq));
      for (int64_t ix = 0; ix < nx; ++ix)
        for (int64_t iq = 0; iq < nq; ++iq) {
          const size_t i = (size_t)(iq * nx + ix);
          if (mask[i]) rows.push_back({(double)period, x_lo + (double)ix * step_, (double)iq * step_, pol[i] * step_});
        }
    }
    return rows;
  }
  Engine& engine() { return engine_; }

  Actions getFeasibleActions;
  Trans stateTransition;
  Imm immediateValue;

 private:
  static sdpgpu_desc desc_of(const LeadtimeFunctor& f) {
    sdpgpu_desc d = make_desc(OptDirection::MIN);
    f.fill(d);
    return d;
  }
  double step_;
  Engine engine_;
};

// ---- sdp.cash.CashRecursion ----------------------------------------------------------------------------
class CashRecursion {
 public:
  using State = cash::CashState;
  using Trans = StateTransitionFunction<State, double, double, State>;
  using Imm = ImmediateValueFunction<State, double, double, double>;
  using Actions = std::function<std::vector<double>(const State&)>;

  CashRecursion(OptDirection optDirection, const Pmf& pmf, Actions getFeasibleAction, Trans stateTransition,
                Imm immediateValue, double discountFactor, const CashFunctor& functor)
      : getFeasibleActions(std::move(getFeasibleAction)), stateTransition(std::move(stateTransition)),
        immediateValue(std::move(immediateValue)), step_(functor.stepSize),
        engine_(desc_of(optDirection, discountFactor, functor), pmf, functor.overheadCosts, functor.user) {}

  Trans getStateTransitionFunction() const { return stateTransition; }
  Imm getImmediateValueFunction() const { return immediateValue; }
  void setTreeMapCacheAction() {}

  double getExpectedValue(const State& s) {
    return engine_.lookup(s.getPeriod(), s.getIniInventory(), s.getIniCash(), 0).first;
  }
  double getAction(const State& s) {
    return engine_.lookup(s.getPeriod(), s.getIniInventory(), s.getIniCash(), 0).second * step_;
  }
  /** rows {period, inventory, cash, Q} (CashRecursion.java:209-218), ordered by (period, inventory, cash). */
  std::vector<std::array<double, 4>> getOptTable() {
    std::vector<std::array<double, 4>> rows;
    for (int period = 1; period <= engine_.periods(); ++period) {
      const auto mask = engine_.reachable(period);
      const auto& pol = engine_.policy(period);
      double x_lo;
      int64_t nx, nc, nq;
      engine_.check(sdpgpu_grid(engine_.handle(), period, &x_lo, &nx, &nc, &nq));
      for (size_t i = 0; i < mask.size(); ++i)
        if (mask[i])
          rows.push_back({(double)period, x_lo + (double)(i / (size_t)nc) * step_,
                          sdpgpu_cash_value(engine_.handle(), (int64_t)(i % (size_t)nc)), pol[i] * step_});
    }
    return rows;
  }
  Engine& engine() { return engine_; }

  Actions getFeasibleActions;
  Trans stateTransition;
  Imm immediateValue;

 private:
  static sdpgpu_desc desc_of(OptDirection dir, double discountFactor, const CashFunctor& f) {
    sdpgpu_desc d = make_desc(dir);
    f.fill(d);
    d.discount_factor = discountFactor;
    return d;
  }
  double step_;
  Engine engine_;
};

// ---- sdp.cash.RiskRecursion (survival probability, RiskRecursion.java:31-46, :65-108) -----------------
class RiskRecursion {
 public:
  using State = cash::RiskState;
  using Trans = StateTransitionFunction<State, double, double, State>;
  using Imm = ImmediateValueFunction<State, double, double, double>;
  using Actions = std::function<std::vector<double>(const State&)>;

  RiskRecursion(const Pmf& pmf, Actions getFeasibleAction, Trans stateTransition, Imm immediateValue,
                const SurvivalFunctor& functor)
      : getFeasibleActions(std::move(getFeasibleAction)), stateTransition(std::move(stateTransition)),
        immediateValue(std::move(immediateValue)), step_(functor.stepSize),
        engine_(desc_of(functor), pmf, functor.overheadCosts, functor.user) {}

  Trans getStateTransitionFunction() const { return stateTransition; }
  Imm getImmediateValueFunction() const { return immediateValue; }
  void setTreeMapCacheAction() {}

  double getSurvProb(const State& s) {
    return engine_.lookup(s.getPeriod(), s.getIniInventory(), s.getIniCash(), 0).first;
  }
  double getAction(const State& s) {
    return engine_.lookup(s.getPeriod(), s.getIniInventory(), s.getIniCash(), 0).second * step_;
  }
  /** rows {period, inventory, cash, bankruptBefore (always 0), Q} (RiskRecursion.java:123-132). */
  std::vector<std::array<double, 5>> getOptTable() {
    std::vector<std::array<double, 5>> rows;
    for (int period = 1; period <= engine_.periods(); ++period) {
      const auto mask = engine_.reachable(period);
      const auto& pol = engine_.policy(period);
      double x_lo;
      int64_t nx, nc, nq;
      engine_.check(sdpgpu_grid(engine_.handle(), period, &x_lo, &nx, &nc, &nq));
      for (size_t i = 0; i < mask.size(); ++i)
        if (mask[i])
          rows.push_back({(double)period, x_lo + (double)(i / (size_t)nc) * step_,
                          sdpgpu_cash_value(engine_.handle(), (int64_t)(i % (size_t)nc)), 0.0, pol[i] * step_});
    }
    return rows;
  }
  Engine& engine() { return engine_; }

  Actions getFeasibleActions;
  Trans stateTransition;
  Imm immediateValue;

 private:
  static sdpgpu_desc desc_of(const SurvivalFunctor& f) {
    sdpgpu_desc d = make_desc(OptDirection::MAX);
    f.fill(d);
    d.discount_factor = 1;
    return d;
  }
  double step_;
  Engine engine_;
};

// ---- sdp.cash.CashLeadtimeRecursion (MAX only, CashLeadtimeRecursion.java:53,70) -----------------------
class CashLeadtimeRecursion {
 public:
  using State = cash::CashLeadtimeState;
  using Trans = StateTransitionFunction<State, double, double, State>;
  using Imm = ImmediateValueFunction<State, double, double, double>;
  using Actions = std::function<std::vector<double>(const State&)>;

  CashLeadtimeRecursion(const Pmf& pmf, Actions getFeasibleAction, Trans stateTransition, Imm immediateValue,
                        const CashLeadtimeFunctor& functor)
      : getFeasibleActions(std::move(getFeasibleAction)), stateTransition(std::move(stateTransition)),
        immediateValue(std::move(immediateValue)), step_(functor.stepSize),
        engine_(desc_of(functor), pmf, functor.overheadCosts, functor.user) {}

  double getExpectedValue(const State& s) {
    return engine_.lookup(s.getPeriod(), s.getIniInventory(), s.getIniCash(), s.getPreQ()).first;
  }
  double getAction(const State& s) {
    return engine_.lookup(s.getPeriod(), s.getIniInventory(), s.getIniCash(), s.getPreQ()).second * step_;
  }
  Engine& engine() { return engine_; }

  Actions getFeasibleActions;
  Trans stateTransition;
  Imm immediateValue;

 private:
  static sdpgpu_desc desc_of(const CashLeadtimeFunctor& f) {
    sdpgpu_desc d = make_desc(OptDirection::MAX);
    f.fill(d);
    return d;
  }
  double step_;
  Engine engine_;
};

// ---- workforce.StaffRecursion (StaffRecursion.java:42-118, 237-254): MIN only ---------------------------
struct StaffFunctor {  // WorkforcePlanning.java:72-101 (clampStaff) / WorkforceTesting.java:80-107 (no clamp)
  double fixCost = 0, unitVariCost = 0, salary = 0, unitPenalty = 0;
  std::vector<int> minStaffNum;
  int maxHireNum = 0, minX = 0, maxX = 0, iniStaffNum = 0;
  bool clampStaff = true;
  void fill(sdpgpu_desc& d) const {
    d.family = SDPGPU_FAMILY_STAFF;
    d.step = 1;
    d.min_inventory = minX;
    d.max_inventory = maxX;
    d.clamp_inventory = clampStaff ? 1 : 0;
    d.ini_inventory = iniStaffNum;
    d.max_order_quantity = maxHireNum;
    d.fixed_order_cost = fixCost;
    d.unit_order_cost = unitVariCost;
    d.holding_cost = salary;
    d.penalty_cost = unitPenalty;
  }
};

class StaffRecursion {
 public:
  using State = workforce::StaffState;
  using Trans = StateTransitionFunction<State, int, int, State>;
  using Imm = ImmediateValueFunction<State, int, int, double>;
  using Actions = std::function<std::vector<int>(const State&)>;

  StaffRecursion(Actions getFeasibleAction, Trans stateTransition, Imm immediateValue, const LevelPmf& pmf, int T,
                 const StaffFunctor& functor)
      : getFeasibleAction(std::move(getFeasibleAction)), stateTransition(std::move(stateTransition)),
        immediateValue(std::move(immediateValue)), engine_(desc_of(functor), pmf, functor.minStaffNum) {
    if (T != (int)pmf.size()) throw std::invalid_argument("T != pmf.length");
  }

  Trans getStateTransitionFunction() const { return stateTransition; }
  Imm getImmediateValueFunction() const { return immediateValue; }

  double getExpectedValue(const State& s) { return at(s).first; }
  int getAction(const State& s) { return at(s).second; }
  /** rows {period, iniStaffNum, action} of the visited states (StaffRecursion.java:245-254). */
  std::vector<std::array<double, 3>> getOptTable() {
    std::vector<std::array<double, 3>> rows;
    for (int period = 1; period <= engine_.periods(); ++period) {
      const auto mask = engine_.reachable(period);
      const auto& pol = engine_.policy(period);
      double x_lo;
      int64_t nx, nc, nq;
      engine_.check(sdpgpu_grid(engine_.handle(), period, &x_lo, &nx, &nc, &nq));
      for (size_t i = 0; i < mask.size(); ++i)
        if (mask[i]) rows.push_back({(double)period, x_lo + (double)i, (double)pol[i]});
    }
    return rows;
  }
  Engine& engine() { return engine_; }

  Actions getFeasibleAction;
  Trans stateTransition;
  Imm immediateValue;

 private:
  std::pair<double, int> at(const State& s) {
    engine_.solve();
    if (sdpgpu_state_index(engine_.handle(), s.period, (double)s.iniStaffNum, 0, 0) < 0)
      throw std::out_of_range("staff number outside what the recursion can reach");  // the reference would NPE
    return engine_.lookup(s.period, (double)s.iniStaffNum, 0, 0);
  }
  static sdpgpu_desc desc_of(const StaffFunctor& f) {
    sdpgpu_desc d = make_desc(OptDirection::MIN);
    f.fill(d);
    return d;
  }
  Engine engine_;
};

// ---- sdp.cash.multiItem.CashRecursionMulti / CashRecursionMultiXR (two products, cash-constrained) -------------
// CashRecursionMulti.java:39-57,82-116 over the lambdas of MultiItemCash.java:66-118 (model 1) and
// CashRecursionMultiXR.java:39-57,60-96 over those of MultiItemCashXR.java:92-148 (model 2).  The lambdas are fixed
// in form, so the functor carries their parameters; the joint pmf of period t is the list GetPmfMulti.getPmf(t)
// returns: rows {d1, d2, probability}.
struct MultiItemFunctor {
  int Qbound = 0;
  std::array<double, 2> price{}, variCost{}, salPrice{};
  double minInventoryState = 0, maxInventoryState = 0, minCashState = 0, maxCashState = 0, depositeRate = 0;
};
using MultiPmf = std::vector<std::vector<std::array<double, 3>>>;

namespace multiItem {
struct CashStateMulti {  // CashStateMulti.java:14-70; for the XR class `iniCash` holds R (CashStateMultiXR.java:21-73)
  int period;
  double iniInventory1, iniInventory2, iniCash;
  bool operator<(const CashStateMulti& o) const {
    return std::tie(period, iniInventory1, iniInventory2, iniCash) < std::tie(o.period, o.iniInventory1, o.iniInventory2, o.iniCash);
  }
};
}  // namespace multiItem

template <int MODEL>
class CashRecursionMultiBase {
 public:
  using State = multiItem::CashStateMulti;
  CashRecursionMultiBase(double discountFactor, MultiPmf pmf, int TLength, MultiItemFunctor functor)
      : discount_(discountFactor), pmf_(std::move(pmf)), T_(TLength), f_(functor) {}

  /** The first call (a period-1 state) solves and reads the whole memo back; later calls answer from it. */
  double getExpectedValue(const State& s) { return find(s)[0]; }
  /** Actions (Q1, Q2) for CashRecursionMulti, order-up-to levels (y1, y2) for CashRecursionMultiXR. */
  std::array<int, 2> getAction(const State& s) {
    const auto& e = find(s);
    return {(int)e[1], (int)e[2]};
  }
  const std::map<State, std::array<double, 3>>& getCacheActions() const { return memo_; }  // key order = the reference's

 private:
  const std::array<double, 3>& find(const State& s) {
    if (memo_.empty()) solve(s);
    auto it = memo_.find(s);
    if (it == memo_.end()) throw std::out_of_range("state was not visited from the initial state");
    return it->second;
  }
  void solve(const State& ini) {
    if (ini.period != 1) throw std::invalid_argument("the first query fixes the initial (period-1) state");
    std::vector<int32_t> off{0};
    std::vector<double> d1, d2, p;
    for (const auto& tile : pmf_) {
      for (const auto& r : tile) {
        d1.push_back(r[0]);
        d2.push_back(r[1]);
        p.push_back(r[2]);
      }
      off.push_back((int32_t)d1.size());
    }
    sdpgpu_multicash k{};
    k.T = T_;
    k.q_bound = f_.Qbound;
    for (int i = 0; i < 2; ++i) {
      k.price[i] = f_.price[i];
      k.vari_cost[i] = f_.variCost[i];
      k.sal_price[i] = f_.salPrice[i];
    }
    k.ini_cash = ini.iniCash;
    k.ini_i1 = ini.iniInventory1;
    k.ini_i2 = ini.iniInventory2;
    k.min_inventory = f_.minInventoryState;
    k.max_inventory = f_.maxInventoryState;
    k.min_cash = f_.minCashState;
    k.max_cash = f_.maxCashState;
    k.discount = discount_;
    k.pmf_off = off.data();
    k.d1 = d1.data();
    k.d2 = d2.data();
    k.p = p.data();
    auto run = [&](int64_t* states) {
      double fv, ms;
      int32_t a1, a2;
      int64_t cells;
      const int rc = MODEL == 2 ? sdpgpu_multixr_solve(&k, f_.depositeRate, &fv, &a1, &a2, states, &cells, &ms)
                                : sdpgpu_multicash_solve(&k, &fv, &a1, &a2, states, &cells, &ms);
      if (rc != 0) throw std::runtime_error(sdpgpu_multilead_last_error());
    };
    std::vector<int64_t> states((size_t)T_);
    run(states.data());
    int64_t rows = 0;
    for (int64_t n : states) rows += n;
    std::vector<int32_t> period((size_t)rows), a1((size_t)rows), a2((size_t)rows);
    std::vector<double> i1((size_t)rows), i2((size_t)rows), q1((size_t)rows), q2((size_t)rows), cash((size_t)rows), value((size_t)rows);
    sdpgpu_multi_table tab{rows, 0, period.data(), i1.data(), i2.data(), q1.data(), q2.data(), cash.data(), value.data(), a1.data(), a2.data()};
    sdpgpu_multi_set_table(&tab);
    try {
      run(states.data());
    } catch (...) {
      sdpgpu_multi_set_table(nullptr);
      throw;
    }
    sdpgpu_multi_set_table(nullptr);
    for (int64_t r = 0; r < tab.rows; ++r)
      memo_[State{period[(size_t)r], i1[(size_t)r], i2[(size_t)r], cash[(size_t)r]}] = {value[(size_t)r], (double)a1[(size_t)r], (double)a2[(size_t)r]};
  }
  double discount_;
  MultiPmf pmf_;
  int T_;
  MultiItemFunctor f_;
  std::map<State, std::array<double, 3>> memo_;
};
using CashRecursionMulti = CashRecursionMultiBase<1>;
using CashRecursionMultiXR = CashRecursionMultiBase<2>;

}  // namespace gpu
}  // namespace sdp
