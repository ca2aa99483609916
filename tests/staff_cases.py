"""Small instances of the STAFF family (workforce.StaffRecursion), shared by the CPU and GPU tests."""
import numpy as np

from stochastic_inventory_amd.pmf import staff_level_pmf
from stochastic_inventory_amd.workforce import StaffFunctor


class StaffCase:
    def __init__(self, name, functor, table, row_len=None):
        self.name, self.functor, self.table, self.row_len = name, functor, np.ascontiguousarray(table), row_len
        self.T = self.table.shape[0]

    def oracle_problem(self, staffref):
        f = self.functor
        return staffref.Problem(T=self.T, min_x=f.minX, max_x=f.maxX, clamp=f.clampStaff, ini_x=f.iniStaffNum,
                                max_hire=f.maxHireNum, fix_cost=f.fixCost, unit_vari_cost=f.unitVariCost,
                                salary=f.salary, unit_penalty=f.unitPenalty, min_staff=list(f.minStaffNum),
                                prob=self.table, row_len=self.row_len)


def staff_planning_small(T=3):
    """WorkforcePlanning.java:33-50 scaled down: clamped staff numbers, one turnover rate."""
    f = StaffFunctor(fixCost=100, unitVariCost=10, salary=20, unitPenalty=80, minStaffNum=[8, 8, 8][:T], maxHireNum=20,
                     minX=0, maxX=30, clampStaff=True, iniStaffNum=0)
    return StaffCase("staff_planning_small", f, staff_level_pmf([0.5] * T, 31))


def staff_testing_small(T=4):
    """WorkforceTesting.java:57-107 scaled down: no clamp (the grid grows by maxHireNum a period), a table shorter
    than the staff numbers reached (levels beyond it use its last row), varying minimum staff."""
    f = StaffFunctor(fixCost=50, unitVariCost=20, salary=5, unitPenalty=250, minStaffNum=[4, 9, 6, 3][:T],
                     maxHireNum=12, clampStaff=False, iniStaffNum=0)
    return StaffCase("staff_testing_small", f, staff_level_pmf([0.3] * T, 13))


def staff_rates(T=3):
    """A different turnover rate every period, a start above zero, fractional costs."""
    f = StaffFunctor(fixCost=12.5, unitVariCost=1.25, salary=2.75, unitPenalty=9.5, minStaffNum=[5, 2, 7][:T],
                     maxHireNum=9, minX=0, maxX=40, clampStaff=True, iniStaffNum=3)
    return StaffCase("staff_rates", f, staff_level_pmf([0.1, 0.6, 0.35][:T], 41))


def staff_short_rows(T=3):
    """Rows shorter than y + 1 (a truncated turnover distribution): exercises row_len."""
    rows = 25
    full = staff_level_pmf([0.4] * T, rows)
    row_len = np.minimum(np.arange(rows) + 1, 6).astype(np.int32)
    table = np.zeros((T, rows, 6))
    for y in range(rows):
        n = row_len[y]
        table[:, y, :n] = full[:, y, :n] / full[:, y, :n].sum(axis=1, keepdims=True)
    f = StaffFunctor(fixCost=30, unitVariCost=3, salary=4, unitPenalty=40, minStaffNum=[6] * T, maxHireNum=10,
                     minX=0, maxX=24, clampStaff=True, iniStaffNum=2)
    return StaffCase("staff_short_rows", f, table, row_len)


def staff_wide_actions(T=2):
    """Few states, many actions: the period runs as many action groups per state tile (partial rows + combine)."""
    f = StaffFunctor(fixCost=100, unitVariCost=10, salary=20, unitPenalty=80, minStaffNum=[40] * T, maxHireNum=300,
                     minX=0, maxX=90, clampStaff=True, iniStaffNum=0)
    return StaffCase("staff_wide_actions", f, staff_level_pmf([0.5] * T, 91))


def staff_single(T=1):
    """Degenerate sizes: one period, one action, one state."""
    f = StaffFunctor(fixCost=1, unitVariCost=1, salary=1, unitPenalty=2, minStaffNum=[1], maxHireNum=0, minX=2, maxX=2,
                     clampStaff=True, iniStaffNum=2)
    return StaffCase("staff_single", f, staff_level_pmf([0.5], 3))


ALL = [staff_planning_small, staff_testing_small, staff_rates, staff_short_rows, staff_wide_actions, staff_single]
