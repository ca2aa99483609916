"""Pure-Python literal translation of the reference's memoised recursion (Recursion.java:89-163,
CashRecursion.java:79-140) over the host-side functor restatements.  Small cases only; a third,
independent statement of the loop used to cross-check the C oracle."""
import sys

from stochastic_inventory_amd.states import OptDirection

DBL_MAX = sys.float_info.max


def memo_recursion(functor, pmf, direction, ini_state, cash_loop=False, gamma=1.0):
    T = len(pmf)
    cache_values, cache_actions = {}, {}

    def get_expected_value(s):
        if s in cache_values:
            return cache_values[s]
        feasible = functor.feasibleActions(s, T)
        d_and_p = pmf[s.getPeriod() - 1]
        val = DBL_MAX if direction == OptDirection.MIN else -DBL_MAX
        best = 0.0
        for order_qty in feasible:
            q = 0.0
            for dp in d_and_p:
                d, p = float(dp[0]), float(dp[1])
                if cash_loop:
                    this_d = functor.immediateValue(s, order_qty, d, T)
                    q += p * this_d
                    if s.getPeriod() < T:
                        ns = functor.stateTransition(s, order_qty, d, T)
                        q += p * gamma * get_expected_value(ns)
                else:
                    q += p * functor.immediateValue(s, order_qty, d, T)
                    if s.getPeriod() < T:
                        ns = functor.stateTransition(s, order_qty, d, T)
                        q += p * get_expected_value(ns)
            if direction == OptDirection.MIN:
                if q < val:
                    val, best = q, order_qty
            else:
                if q > val:
                    val, best = q, order_qty
        cache_values[s] = val
        cache_actions[s] = best
        return val

    root = get_expected_value(ini_state)
    return root, cache_values, cache_actions


def surv_recursion(functor, pmf, ini_state, gamma=1.0):
    """RiskRecursion.getSurvProb (RiskRecursion.java:65-108; CashRecursion.java:143-194 with the discount)."""
    T = len(pmf)
    cache_values, cache_actions = {}, {}

    def get_surv_prob(s):
        if s in cache_values:
            return cache_values[s]
        feasible = functor.feasibleActions(s, T)
        d_and_p = pmf[s.getPeriod() - 1]
        val, best = -DBL_MAX, 0.0
        for order_qty in feasible:
            q = 0.0
            for dp in d_and_p:
                d, p = float(dp[0]), float(dp[1])
                if s.getPeriod() == T:
                    final_cash = s.getIniCash() + functor.immediateValue(s, order_qty, d, T)
                    q += p * (1 if final_cash >= 0 else 0)
                if s.getPeriod() < T:
                    ns = functor.stateTransition(s, order_qty, d, T)
                    this_prob = 0 if ns.getIniCash() < 0 else get_surv_prob(ns)
                    q += p * gamma * this_prob
            if q > val:
                val, best = q, order_qty
        cache_values[s] = val
        cache_actions[s] = best
        return val

    root = get_surv_prob(ini_state)
    return root, cache_values, cache_actions


def staff_recursion(functor, table, row_len, T):
    """workforce.StaffRecursion.getExpectedValue (StaffRecursion.java:81-118) as literal memoised Python over the
    functor's host lambdas.  table[t][y][j]: probability of turnover j at hire-up-to level y.
    -> (root value, {(period, staff): (value, action)})"""
    sys.setrecursionlimit(100000)
    from stochastic_inventory_amd.workforce import StaffState
    cache = {}
    n_rows = len(table[0])

    def value(s):
        key = (s.period, s.iniStaffNum)
        if key in cache:
            return cache[key][0]
        t = s.period - 1
        bestHireQty, val = 0, 1.7976931348623157e308
        for orderQty in functor.feasibleActions(s):
            hireUpTo = s.iniStaffNum + orderQty
            if hireUpTo >= n_rows - 1:
                hireUpTo = n_rows - 1
            n = int(row_len[hireUpTo]) if row_len is not None else hireUpTo + 1
            thisQValue = 0.0
            for demand in range(n):
                p = float(table[t][hireUpTo][demand])
                thisQValue += p * functor.immediateValue(s, orderQty, demand)
                if s.period < T:
                    thisQValue += p * value(functor.stateTransition(s, orderQty, demand))
            if thisQValue < val:
                val, bestHireQty = thisQValue, orderQty
        cache[key] = (val, bestHireQty)
        return val

    root = value(StaffState(1, functor.iniStaffNum))
    return root, cache
