/*
 * sdpgpu.h -- C ABI of the MI355X-native finite-horizon SDP engine.
 *
 * This is the drop-in boundary for ONE path of RobinChen121/Stochastic-Inventory:
 * the backward Bellman recursion of src/sdp (the `getExpectedValue` bodies).  A
 * host in any language (the reference's Java over JNI, C++, Python ctypes) binds
 * exactly these entry points.  Plain pointers and sizes only: no JNI, torch or
 * C++ types cross this line, no exception crosses it, every call returns an int
 * status (0 = ok) and the text of the last failure is read with
 * sdpgpu_last_error().
 *
 * Reference interface each entry point replaces (paths relative to the reference
 * checkout):
 *
 *   sdpgpu_create            constructors  src/sdp/inventory/Recursion.java:49-63,
 *                            src/sdp/inventory/LeadtimeRecursion.java:28-45,
 *                            src/sdp/cash/CashRecursion.java:39-56,
 *                            src/sdp/cash/CashLeadtimeRecursion.java:28-46,
 *                            src/capacitated/CLSP.java:22-24.  The three Java lambdas
 *                            (feasible actions / StateTransitionFunction /
 *                            ImmediateValueFunction; StateTransition.java:20-22,
 *                            ImmediateValue.java:23-25) cannot be called from a GPU;
 *                            they are named by `family` + the scalar parameters the
 *                            in-scope drivers close over (see sdpgpu_desc).
 *   sdpgpu_set_pmf           the `double[][][] pmf` constructor argument
 *                            (Recursion.java:38,54): pmf[t][j] = {demand, prob}.
 *                            Also RiskRecursion.java:31-46 (family SURVIVAL).
 *   sdpgpu_solve             the first `getExpectedValue(initialState)` / `getSurvProb(initialState)` call
 *                            (Recursion.java:89-163, CLSP.java:88-138,
 *                            LeadtimeRecursion.java:47-75, CashRecursion.java:79-140,
 *                            CashLeadtimeRecursion.java:48-79): fills the value and
 *                            action maps.  Here: dense backward sweep t = T..1.
 *   sdpgpu_run_period        one level of that recursion (all states of period t).
 *   sdpgpu_solve_sharded /   the same first call with the state axis cut over the GPUs of a node (one rank per
 *   sdpgpu_solve_multi       process, or all devices from one process): still ONE call per rank / per JVM.
 *   sdpgpu_values            `cacheValues` lookups (Recursion.java:36,90).
 *   sdpgpu_policy            `cacheActions` lookups / getAction / getCacheActions
 *                            (Recursion.java:35,160,165-171).
 *   sdpgpu_eval_states       `getExpectedValue(state)` for a state that is not a
 *                            grid point (e.g. an off-grid period-1 initial cash,
 *                            CashConstraint.java:141) and the lazy re-entry of the
 *                            simulators (Simulation.java:62-63).
 *   sdpgpu_simulate          the rollout loops of the simulators (Simulation.java:59-69,
 *                            CashSimulation.java:101-112): policy look-up + imm + transition.
 *   sdpgpu_reachable         the key set of `cacheActions`, i.e. the states the
 *                            memoised recursion would have visited; getOptTable
 *                            (Recursion.java:177-186) lists exactly those.
 *
 * Numerics contract: all arithmetic fp64, no FMA contraction, demand index
 * ascending, `acc += p*imm; acc += p*[gamma*]V(next)` in that order, strict </>
 * arg-opt in ascending action order (lowest action index wins ties), so values
 * are bit-identical to a literal CPU restatement of the Java loops and the policy
 * indices are bit-exact.
 *
 * Threading: a handle is NOT thread-safe; distinct handles are independent.
 * Ownership: the caller owns every host buffer; the library copies what it needs
 * at the call and owns its device tables unless sdpgpu_attach_values is used.
 */
#ifndef SDPGPU_H
#define SDPGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default) /* the library is built with -fvisibility=hidden: only this header is exported */
#endif

#define SDPGPU_ABI_VERSION 6

/* status codes */
#define SDPGPU_OK 0
#define SDPGPU_ERR_ARG 1      /* bad descriptor / argument */
#define SDPGPU_ERR_STATE 2    /* call out of order (e.g. values before solve) */
#define SDPGPU_ERR_DEVICE 3   /* HIP runtime error (message carries hipGetErrorString) */
#define SDPGPU_ERR_UNSUPPORTED 4
#define SDPGPU_ERR_ALLOC 5    /* ABI 6: a host allocation failed (std::bad_alloc caught at the boundary) */
#define SDPGPU_ERR_INTERNAL 6 /* ABI 6: any other C++ exception caught at the boundary -- none crosses it */

/* Functor families = the closed-form lambda families of the in-scope drivers. */
typedef enum sdpgpu_family {
  /* state (x). CLSP.java:251-272, CLSPTesting.java:78-106, CLSPforDraw.java:86-103,
   * LevelFitsS.java:85-102.  Backorder, two-ternary clamp, imm on unclamped level. */
  SDPGPU_FAMILY_BACKORDER = 1,
  /* state (x, preQ). Leadtime.java:50-81.  Lead time 1, order arrives next period.  With
   * desc.lead_time = 2 (a generalisation the reference does not have; BASELINE configs[3]) the state is
   * (x, q1, q2): q1 arrives this period, q2 the next; x' = x + q1 - d, q1' = q2, q2' = action. */
  SDPGPU_FAMILY_LEADTIME = 2,
  /* state (x, cash). CashConstraint.java:95-133 (cash_formula 0) and
   * CashConstraintTesting.java:110-148 (cash_formula 1).  Lost sales, cash-limited orders.
   * cash_formula 2: sdp.cash.CashRecursionXR.getExpectedValue (CashRecursionXR.java:79-126) over the lambdas of
   * cash.singleItem.CashConstraintXR (CashConstraintXR.java:84-125; Chao 2008): state (x, R) with working capital
   * R = cash + variCost * x, actions = order-up-to levels y = x, x + 1, ... up to max(x, R / variCost).  The grid is
   * (inventory, cash) as for formula 0 -- the transition rounds the cash balance (:120) before it forms R (:121) --
   * and wherever this header says `cash` for a state of this family (ini_cash, sdpgpu_state_index,
   * sdpgpu_eval_states) the value is R.  Policy indices k stand for y = x + k (getAction = x + k * step).
   * max_order_quantity is not read (the reference's list is bounded by R / variCost only). */
  SDPGPU_FAMILY_CASH = 3,
  /* state (x, cash). CashOverdraft.java:72-118.  Piecewise overdraft interest. */
  SDPGPU_FAMILY_OVERDRAFT = 4,
  /* state (x, cash, preQ). SingleProductLeadtime.java:72-119. */
  SDPGPU_FAMILY_CASH_LEADTIME = 5,
  /* state (x, cash). The survival-probability recursion: RiskRecursion.getSurvProb (RiskRecursion.java:65-108)
   * == CashRecursion.getSurvProb (CashRecursion.java:143-194, which also multiplies by discount_factor) with the
   * lambdas of cashSurvival.java:98-143: orders limited by cash / variCost, no penalty term.  The value is
   * P(final cash >= 0 and no negative cash on the way) under the best policy, MAX only; a successor with
   * negative cash contributes 0 and is not visited. */
  SDPGPU_FAMILY_SURVIVAL = 6,
  /* state (staff number). workforce.StaffRecursion.getExpectedValue (StaffRecursion.java:81-118) with the lambdas of
   * WorkforcePlanning.java:72-101 (clamp_inventory 1) / WorkforceTesting.java:80-107 (clamp_inventory 0): the pmf of
   * a period depends on the hire-up-to level y = staff + hires (sdpgpu_set_level_pmf, not sdpgpu_set_pmf).  MIN only.
   * Descriptor fields: min/max_inventory = minX/maxX (>= 0), ini_inventory = iniStaffNum, max_order_quantity =
   * maxHireNum, step 1, fixed_order_cost = fixCost, unit_order_cost = unitVariCost, holding_cost = salary,
   * penalty_cost = unitPenalty; minStaffNum[t] goes through sdpgpu_set_overhead (an integer). */
  SDPGPU_FAMILY_STAFF = 7
} sdpgpu_family;

/* OptDirection, Recursion.java:44-47 / CashRecursion.java:34-37. */
typedef enum sdpgpu_direction { SDPGPU_MIN = 0, SDPGPU_MAX = 1 } sdpgpu_direction;

/* kernel selection (0 = let the library choose) */
#define SDPGPU_KERNEL_AUTO 0
#define SDPGPU_KERNEL_GATHER 1 /* generic per-cell functor + gather from V_{t+1} in HBM/L2 */
#define SDPGPU_KERNEL_WINDOW 2 /* F1/F2: LDS-staged {L(l), V(clamp l)} window, register sliding */
#define SDPGPU_KERNEL_SEPARABLE 3 /* OPT-IN, never chosen automatically.
                                     F1: Q(x,a) = c(a) + G(x+a), O((S+A)D + SA) per period.  REASSOCIATES the reference's
                                     sum: values agree to rounding (1e-9 relative), the arg-opt may differ on near-ties.
                                     F2 (lead time 1 or 2): V_t and the arg-min depend on (x + preQ[, q2]) only: one
                                     evaluation per level in the reference's operation order, O(A (nx+nq) D [nq]) for
                                     that table + one write per state.  EXACT: values and policy bit-identical to the
                                     cell-by-cell kernels (tests/test_gpu_separable.py).
                                     F5 (cash + lead time, ABI 6): SingleProductLeadtime's lambdas read x and preQ through
                                     x + preQ only (SingleProductLeadtime.java:82-119), so every row of a level holds the same
                                     tables: one representative row per level through the cash row kernel, cell by cell in the
                                     reference's order, then one copy per state -- (nx + nq - 1) rows evaluated instead of
                                     nx nq.  EXACT; one rank only. */

/*
 * Problem descriptor: everything the reference's lambdas close over.  Field names
 * follow the reference's variable names.  Inventory, action and demand values must
 * be integer multiples of `step` and `step` itself integer-valued (every in-scope
 * driver uses stepSize = 1), so that x + a - d is exact in fp64 as it is in Java.
 */
typedef struct sdpgpu_desc {
  int32_t abi_version; /* SDPGPU_ABI_VERSION */
  int32_t family;      /* sdpgpu_family */
  int32_t direction;   /* sdpgpu_direction */
  int32_t periods;     /* T = pmf.length */

  /* inventory axis */
  double step;               /* stepSize */
  double min_inventory;      /* minState / minInventory / minInventoryState */
  double max_inventory;      /* maxState / maxInventory / maxInventoryState */
  double max_order_quantity; /* maxOrderQuantity: actions 0, step, ...; count (int)(Q/step)+1 for BACKORDER / LEADTIME
                                (`new double[(int)(maxOrderQuantity/stepSize)+1]`, CLSPTesting.java:79), (int)Q+1 WHATEVER the
                                step for the cash families (`iterate(0, i -> i + stepSize).limit((int) maxQ + 1)`,
                                CashConstraint.java:99).  CASH_LEADTIME, whose order becomes the next state's preQ, therefore
                                needs step = 1 (SDPGPU_ERR_UNSUPPORTED otherwise), like cash_formula 2 and STAFF. */
  int32_t clamp_inventory;   /* 1: clamp as CLSP.java:257-258; 0: no clamp (Leadtime.java:65-66
                                has the clamp commented out) -> per-period boxes grown from ini_* */
  int32_t zero_order_last_period; /* SingleProductLeadtime.java:74-75: maxQ = 0 when period == T */

  /* period-1 initial state: bounding boxes of unclamped families, reachable-set filter */
  double ini_inventory;
  double ini_cash;
  double ini_preq;

  /* cost parameters (F1/F2) */
  double fixed_order_cost; /* fixedOrderingCost / fixOrderCost (K) */
  double unit_order_cost;  /* proportionalOrderingCost / variOrderingCost / variCost (v) */
  double holding_cost;     /* holdingCost (h) */
  double penalty_cost;     /* penaltyCost: backorder pi (F1/F2); endCash<0 multiplier (F3) */

  /* cash families (F3/F4/F5) */
  double price;
  double salvage_value;   /* applied only when period == T */
  double deposit_rate;    /* depositeRate (F3 formula 0) */
  double overhead_cost;   /* overheadCost, same every period unless sdpgpu_set_overhead */
  double overhead_rate;   /* overheadRate (F3 formula 0) */
  double discount_factor; /* CashRecursion.java:120: p * gamma * V, evaluated (p*gamma)*V */
  double min_cash;
  double max_cash;
  double cash_round_mult; /* Math.round(nextCash * mult) ... */
  double cash_round_div;  /* ... / div */
  int32_t cash_round_int_div; /* 1: `/ 10` long division (CashOverdraft.java:116); 0: `/ 10.0` */
  int32_t cash_formula;       /* F3: 0 CashConstraint.java:103-119, 1 CashConstraintTesting.java:117-132,
                                 2 CashConstraintXR.java:91-105 (state (x, R), order-up-to actions) */

  /* overdraft interest schedule (F4/F5): CashOverdraft.java:86-95 */
  double r0;
  double r2;
  double r3;
  double overdraft_limit;
  double interest_free_amount;

  /* execution */
  int32_t kernel;      /* SDPGPU_KERNEL_* */
  int32_t device;      /* HIP device ordinal, -1 = current device */
  int32_t rank;        /* state-axis slab owned by this handle: rank of world_size */
  int32_t world_size;  /* 1 = whole grid */
  int32_t store_all_values; /* 1: keep V_t for every t (values query); 0: two ping-pong tables */
  int32_t lead_time;   /* LEADTIME family only: 0 or 1 = the reference's lead time 1 (Leadtime.java); 2 = two-stage
                          pipeline, state (x, q1, q2), needs clamp_inventory = 1 */
  double ini_preq2;    /* lead_time 2: q2 of the period-1 state (ini_preq is q1) */
  double reserved1;
} sdpgpu_desc;

typedef struct sdpgpu_stats {
  int64_t states_total;     /* sum over periods of grid states (whole grid, all ranks) */
  int64_t cells_evaluated;  /* sum over periods and THIS rank's states of nA(s) * D_t */
  int64_t cells_all_ranks;  /* the same over the whole grid */
  double  solve_ms;         /* HIP-event time of the last sdpgpu_solve on its stream */
  double  kernel_ms_sum;    /* sum of per-period kernel times when profiling is on, else 0 */
  int32_t periods_run;
  int32_t kernel_used;      /* SDPGPU_KERNEL_* actually launched for the last period run */
  int32_t window_r;         /* F1 window kernel: actions per register block ... */
  int32_t window_s;         /* ... and adjacent states per lane of the plan used for period 1 (0: other kernel) */
  double  fp64_ops_executed; /* sum over periods of THIS rank's cells x the fp64 add/mul operations the kernel that ran
                                the period executes per cell (the reference's five per cell minus the ones that are the
                                same operation on the same operands in neighbouring cells and are formed once; window
                                and uniform-shift kernels).  0 when a period ran a kernel without such a model. */
  double  lds_bytes;        /* ABI 4: bytes THIS rank's cells moved through the LDS (reads and staging writes) and through the */
  double  l1_bytes;         /* vector L1 (per-cell gathers, staging loads), by the same per-kernel models; 0 where a kernel has
                               none.  bench.py prices the kernels these units bind (cash_diag_kernel: LDS; cash_shift_kernel: L1)
                               against 128 and 64 B/clk/CU. */
  int64_t graph_replays;    /* ABI 5: sdpgpu_solve calls served by ONE hipGraphLaunch of the captured sweep.  Opt-in, environment
                               SDPGPU_GRAPH=1 (the first call runs eagerly, the second is captured while it is enqueued, later
                               calls replay; per-period profiling, user functors and the legacy NULL stream run eagerly).  Off
                               by default: replay measured 0.6-1.6 % SLOWER than the eager sweep on ROCm 7.2. */
} sdpgpu_stats;

typedef struct sdpgpu_handle sdpgpu_handle;

/* Library identity: returns SDPGPU_ABI_VERSION. */
int sdpgpu_abi_version(void);

/* ABI 6.  Identity of the BINARY: the 16-hex-digit digest of the sources it was built from (every file of csrc/, this
 * header and build.py with its flags -- tools/kernel_sha.py: build_source_sha), baked in at build time by build.py.  The
 * shipped libsdpgpu.so is a prebuilt artefact (git-ignored, carried to the GPU box with the tree): __graft_entry__.smoke()
 * and every bench line compare this string with the digest of the tree they run from, and a mismatch fails smoke.
 * "unknown" when the library was compiled by hand without -DSDPGPU_BUILD_ID.  The reference has no counterpart (a JVM
 * runs the classes it was given: Recursion.java has no native half). */
const char* sdpgpu_build_id(void);

/* Fill a descriptor with the defaults the reference drivers use (discount 1,
 * rounding 10/10.0, clamp on, world 1, store all values, kernel auto). */
void sdpgpu_desc_init(sdpgpu_desc* d);

/* Validate the descriptor, lay out the per-period grids, allocate device tables. */
int sdpgpu_create(const sdpgpu_desc* desc, sdpgpu_handle** out);
void sdpgpu_destroy(sdpgpu_handle* h);

/*
 * User-defined lambdas.  The reference injects a problem as three Java lambdas (Recursion.java:49-52):
 * `Function<State, double[]>` feasible actions, StateTransitionFunction (StateTransition.java:20-22) and
 * ImmediateValueFunction (ImmediateValue.java:23-25).  The built-in families cover the in-scope drivers; any
 * other driver's lambdas are handed over here as HIP device source, compiled at this call with hipRTC for gfx950
 * under the library's numerics contract (-ffp-contract=off) around the same loop (accumulation order,
 * strict-compare arg-opt, discount, survival objective) as the built-in kernels.
 *
 * `functor_source` defines exactly these three device functions (the text may define helpers and constants too):
 *
 *   __device__ int    sdp_feasible_count(const sdp_ctx& c, double x, double cash, double preq);
 *       getFeasibleActions.apply(state).length; action k is k * c.step (DoubleStream.iterate(0, i -> i + stepSize))
 *   __device__ double sdp_immediate(const sdp_ctx& c, double x, double cash, double preq, double action, double demand);
 *       immediateValue.apply(state, action, randomDemand)
 *   __device__ void   sdp_transition(const sdp_ctx& c, double x, double cash, double preq, double action, double demand,
 *                                    double& next_x, double& next_cash, double& next_preq);
 *       stateTransition.apply(state, action, randomDemand), clamped and rounded as the Java lambda does it: the
 *       result must be a grid point of the next period, otherwise the next read of results fails with
 *       SDPGPU_ERR_ARG.
 *
 * ABI 5, optional: a text that also says `#define SDP_USER_CELL 1` and defines
 *   __device__ void   sdp_cell(const sdp_ctx& c, double x, double cash, double preq, double action, double demand,
 *                              double& imm, double& next_x, double& next_cash, double& next_preq);
 * gives the period kernel ONE callback per cell: imm = immediateValue.apply(...) and the successor of the same cell.  The
 * reference's cash-type transitions call immediateValue again (CashConstraint.java:125 `nextCash = cash +
 * immediateValue.apply(...)`); as two functions the increment is spelled out twice per cell.  sdp_immediate and
 * sdp_transition are still required (the reachable-set pass and sdpgpu_eval_states' host twins use them); the
 * natural way to write them is as calls of sdp_cell.
 *
 * ABI 6, optional -- the LEVEL SHAPE: a text that says `#define SDP_SHAPE_LEVEL 1` and defines, INSTEAD of the three lambdas,
 *   __device__ double sdp_action_cost(const sdp_ctx& c, double action);
 *   __device__ double sdp_level_cost(const sdp_ctx& c, double level);
 * declares its lambdas to be
 *   immediateValue  = sdp_action_cost(action) + sdp_level_cost(x + action - demand)      (one addition of the two terms),
 *   stateTransition = x + action - demand, clamped to [min_inventory, max_inventory] as the descriptor says (upper, then lower),
 *   feasible actions = every order 0, step, ... max_order_quantity in every state
 * -- the shape of capacitated.CLSP's lambdas (CLSP.java:251-272) and of the period-1-special second Recursion of CLSPforDraw
 * (CLSPforDraw.java:146-169), with any piecewise, period-dependent costs inside the two functions.  desc->family must be
 * BACKORDER.  The library compiles the two functions with hipRTC, lets them tabulate M(level) and c(action) for every period
 * once (kernel sdp_custom_tabulate), and runs its F1 window kernel from the tables: arbitrary cost closures at the built-in
 * family's speed, and -- each table entry being the user's function of the very level / action the reference would hand it,
 * joined by the one declared addition -- the tables' values are the reference's doubles.  kernel = SDPGPU_KERNEL_GATHER (and
 * sdpgpu_eval_states, sdpgpu_reachable) run the same text through the generic loop; the library forms the three lambdas from
 * the two functions as stated above.
 *
 * with `struct sdp_ctx { int period; int T; double step; const double* params; }` (period = state.getPeriod(),
 * params = the n_params doubles given here: the constants the Java lambdas close over) and the helpers
 * sdp_max / sdp_min / sdp_round / sdp_trunc (java.lang.Math.max / min / round and the (int) cast) and, ABI 6, sdp_ldiv(a, b)
 * (Java's `long / int` on an integer-valued double below 2^31, e.g. `Math.round(cash * 10) / 10`, CashOverdraft.java:116: truncating
 * integer division -- a few integer instructions with a literal divisor, where sdp_trunc(a / b) is an fp64 division per cell).
 *
 * `desc->family` selects the STATE SHAPE and the loop, not the formulas: BACKORDER = (x) under Recursion,
 * LEADTIME = (x, preQ) under LeadtimeRecursion, CASH / OVERDRAFT = (x, cash) under CashRecursion (discounted),
 * CASH_LEADTIME = (x, cash, preQ), SURVIVAL = (x, cash) under getSurvProb.  The grid fields of the descriptor
 * (step, inventory bounds and clamp flag, cash bounds and rounding, max_order_quantity = longest pipeline
 * quantity, ini_*) keep their meaning; the cost fields are ignored.  sdpgpu_simulate is not available (the
 * lambdas also exist on the host, where the reference's simulators call them).
 *
 * How the text is compiled: -O3, no contraction, no fast-math.  On a CLAMPED grid with finite constants the grid, the step and
 * params[] are baked into the generated source as exact literals (SDPGPU_CUSTOM_BAKE=0: read at run time, as before ABI 6), and
 * that compile additionally assumes the lambdas' arithmetic produces no NaN (-fno-honor-nans; SDPGPU_CUSTOM_NNAN=0 turns it
 * off): a clamp the Java source spells `x > c ? c : x` against a non-zero constant then costs one v_min_f64 instead of a compare
 * and two selects.  For lambdas whose values are numbers the results are the same doubles either way
 * (tests/test_gpu_custom_functor.py runs the three modes against each other); what a lambda that does produce a NaN returns is
 * unspecified in the default mode.
 */
int sdpgpu_create_custom(const sdpgpu_desc* desc, const char* functor_source, const double* params, int32_t n_params,
                         sdpgpu_handle** out);

/* Never NULL; empty string when the last call on h succeeded.  h may be NULL to
 * read the message of a failed sdpgpu_create. */
const char* sdpgpu_last_error(const sdpgpu_handle* h);

/* pmf[t] for t = 0..T-1 (period t+1): n pairs, demand[j] ascending as in GetPmf.java:119. */
int sdpgpu_set_pmf(sdpgpu_handle* h, int32_t t, const double* demand, const double* prob, int32_t n);

/* STAFF family: the turnover pmf of period t+1 per hire-up-to level, pmfs[t] of StaffRecursion.java:23,92-95.
 * prob[y * row_stride + j] = pmfs[t][y][j][1] for j < row_len[y] (the realisation pmfs[t][y][j][0] is j itself,
 * WorkforcePlanning.java:57-68); row_len NULL means y + 1 entries per row; a level beyond the table uses its last
 * row.  1 <= row_len[y] <= min(y + 1, row_stride): a realisation never exceeds the staff it applies to. */
int sdpgpu_set_level_pmf(sdpgpu_handle* h, int32_t t, const double* prob, const int32_t* row_len, int32_t n_rows,
                         int32_t row_stride);

/* ---- PMF construction: what a driver runs immediately before the recursion ---------------------
 * `new GetPmf(distributions, truncationQuantile, stepSize).getpmf()[t]` (sdp/inventory/GetPmf.java:82-134; variant
 * SDPGPU_PMF_GETPMF) and capacitated.CLSP.main's inline variant (CLSP.java:219-247; SDPGPU_PMF_CLSP), for the
 * distributions the in-scope drivers use.  GetPmf's structure and quirks are reproduced exactly ((int) truncation of the
 * quantiles, lower bound 0 for integer distributions, prob(j) indexed by position, covered mass as normaliser); the
 * cdf / quantile functions are this library's own double-precision implementations, not SSJ's (parity unpinned at this
 * boundary: the reference records no PMF).  Host arithmetic, no GPU needed.
 * Call with capacity = 0 to learn the number of points of period t (n_out), then with arrays of that size; feed the
 * result to sdpgpu_set_pmf.  Errors through sdpgpu_last_error(NULL). */
#define SDPGPU_DIST_POISSON 1      /* PoissonDist(lambda = a) */
#define SDPGPU_DIST_NORMAL 2       /* NormalDist(mu = a, sigma = b) */
#define SDPGPU_DIST_UNIFORM_INT 3  /* UniformIntDist(i = a, j = b) */
#define SDPGPU_DIST_GAMMA 4        /* GammaDist(alpha = a, lambda = b): shape, RATE (CashConstraintXR.java:67) */
#define SDPGPU_PMF_GETPMF 0
#define SDPGPU_PMF_CLSP 1
typedef struct sdpgpu_dist_spec {
  int32_t kind;
  int32_t reserved;
  double a, b;
} sdpgpu_dist_spec;
int sdpgpu_getpmf(const sdpgpu_dist_spec* distributions, int32_t T, double truncation_quantile, double step,
                  int32_t variant, int32_t t, double* demand, double* prob, int32_t capacity, int32_t* n_out);

/* Optional per-period overhead cost (CashOverdraft.java:38-39 keeps an array).  STAFF family: minStaffNum[t]. */
int sdpgpu_set_overhead(sdpgpu_handle* h, int32_t t, double overhead_cost);

/* Optional: the caller's own action-list LENGTHS.  The reference's `Function<State, double[]> getFeasibleAction`
 * (Recursion.java:49,129) may return any array; every in-scope driver returns a prefix 0, step, 2 step, ... of the
 * action grid whose length follows one of the families' closed forms.  A driver whose list is still such a prefix but
 * with a length of its own (a storage capacity Q <= cap - x, a budget table, ...) hands the lengths over here:
 * counts[i] = getFeasibleAction.apply(state i of period t+1).length for every grid state i (flat index order,
 * n = sdpgpu_num_states), 0 <= counts[i] <= the family's longest list (see max_order_quantity); 0 = no feasible action (the value is
 * then +-Double.MAX_VALUE and the action 0, Recursion.java:132-134).  Such a period runs on the generic kernel (the
 * specialised kernels build on the family's rule).  sdpgpu_eval_states ALWAYS applies the family's rule -- to off-grid
 * states (the caller's list is defined on grid states only) and to on-grid states alike, so for a grid state whose count was
 * overridden here its answer may differ from sdpgpu_values / sdpgpu_policy of the same state: read those for grid states.
 * Before the first run.  Lists that are not prefixes of the action grid need sdpgpu_create_custom. */
int sdpgpu_set_action_counts(sdpgpu_handle* h, int32_t t, const int32_t* counts, int64_t n);

/* Launch kernels on a caller-owned hipStream_t (NULL = the legacy default stream).  Without this
 * call the library creates a non-blocking stream of its own. */
int sdpgpu_set_stream(sdpgpu_handle* h, void* hip_stream);

/* Record a HIP event pair around every period kernel (read back through sdpgpu_period_ms). */
int sdpgpu_set_profiling(sdpgpu_handle* h, int32_t on);

/* ---- geometry ---------------------------------------------------------------------------- */
/* Number of grid states of period t (1-based period, 1..T). */
int64_t sdpgpu_num_states(const sdpgpu_handle* h, int32_t period);
/* Padded row length of the V_t table (multiple of world_size) and this rank's slab [lo, hi). */
int sdpgpu_slab(const sdpgpu_handle* h, int32_t period, int64_t* padded, int64_t* lo, int64_t* hi);
/* Grid of period t: x = x_lo + i*step (i < nx); cash index ic < nc; preQ index iq < nq.
 * Flat index = (iq * nx + ix) * nc + ic.  With lead_time 2 the pipeline axis is the pair
 * iq = iq2 * nq1 + iq1 and sdpgpu_grid reports nq = nq1 * nq2; sdpgpu_grid2 reports both. */
int sdpgpu_grid(const sdpgpu_handle* h, int32_t period, double* x_lo, int64_t* nx, int64_t* nc, int64_t* nq);
int sdpgpu_grid2(const sdpgpu_handle* h, int32_t period, double* x_lo, int64_t* nx, int64_t* nc, int64_t* nq1,
                 int64_t* nq2);
/* Cash value of cash index ic (k / div, exactly the double the reference's rounding yields). */
double sdpgpu_cash_value(const sdpgpu_handle* h, int64_t ic);
/* Dense index of a state tuple in period t, or -1 if it is not a grid point. */
int64_t sdpgpu_state_index(const sdpgpu_handle* h, int32_t period, double x, double cash, double preq);
/* The same with the second pipeline quantity (lead_time 2; preq2 must be 0 otherwise). */
int64_t sdpgpu_state_index2(const sdpgpu_handle* h, int32_t period, double x, double cash, double preq, double preq2);

/* ---- solving ----------------------------------------------------------------------------- */
/* Whole backward sweep t = T..1 on this handle (world_size must be 1).  Asynchronous on the
 * handle's stream unless `sync` != 0. */
int sdpgpu_solve(sdpgpu_handle* h, int32_t sync);
/* One period for this rank's slab, reading the FULL V_{period+1} table.  With world_size > 1
 * the caller all-gathers V_period (device pointer below) across ranks before the next call. */
int sdpgpu_run_period(sdpgpu_handle* h, int32_t period);
/* The same in two halves, so that a sharded caller can overlap the all-gather of V_{period+1} with
 * compute: INTERIOR = the states all of whose cells read only THIS rank's slab of V_{period+1} (it may
 * run while the exchange is in flight; empty for families without a bounded footprint), BOUNDARY = the
 * rest (after the exchange).  INTERIOR followed by BOUNDARY equals sdpgpu_run_period. */
#define SDPGPU_PART_ALL 0
#define SDPGPU_PART_INTERIOR 1
#define SDPGPU_PART_BOUNDARY 2
int sdpgpu_run_period_part(sdpgpu_handle* h, int32_t period, int32_t part);
/* Fewer exchanges on small slabs: the backorder family on the window kernel reads a BOUNDED neighbourhood of
 * V_{period+1} -- state i needs [i - left, i + right] (sdpgpu_footprint; SDPGPU_ERR_UNSUPPORTED when the family has
 * no such bound).  A sharded caller can therefore run K periods between two exchanges by computing period t on its
 * slab widened by the footprints of the periods still to come in the block (sdpgpu_run_period_range; the widening is
 * redundant work that reproduces the neighbours' values bit for bit), and all-gather only to publish the rows.
 * sdpgpu_set_halo (before the first run) announces the largest widening so that the scratch rows are sized for
 * it.  Values are written for the whole range, policy indices only for this rank's own slab. */
int sdpgpu_footprint(const sdpgpu_handle* h, int32_t period, int64_t* left, int64_t* right);
int sdpgpu_set_halo(sdpgpu_handle* h, int64_t halo);
int sdpgpu_run_period_range(sdpgpu_handle* h, int32_t period, int64_t lo, int64_t hi);
/* Device address of the V_period table (padded row, fp64) -- for the in-place all-gather. */
void* sdpgpu_values_device_ptr(sdpgpu_handle* h, int32_t period);
/* Use caller-owned device memory for the value tables: `bytes` >= sdpgpu_values_bytes(h). */
size_t sdpgpu_values_bytes(const sdpgpu_handle* h);
int sdpgpu_attach_values(sdpgpu_handle* h, void* device_ptr, size_t bytes);
/* The row a sharded caller all-gathers after sdpgpu_run_period(period) and before the next period:
 * padded_row 8-byte elements (sdpgpu_slab).  Usually the fp64 V_period row; on small grids, where a
 * tile is shared by several tasks, the row of order-preserving uint64 keys the kernels reduce into and
 * the next period reads directly (the fp64 row is then written by sdpgpu_finalize).  The key arena can be
 * caller memory, like the value arena: sdpgpu_keys_bytes() is 0 when the handle never uses keys. */
void* sdpgpu_exchange_ptr(sdpgpu_handle* h, int32_t period);
size_t sdpgpu_keys_bytes(const sdpgpu_handle* h);
int sdpgpu_attach_keys(sdpgpu_handle* h, void* device_ptr, size_t bytes);

/* Enqueue (do not wait for) whatever deferred read-out work is outstanding, so that in stream order every
 * V_t and policy row of the periods run so far is complete.  On small grids the window kernel leaves
 * the arg-opt of a period as per-chunk rows and resolves all periods in one launch; every reader
 * below does this implicitly, a caller that reads the device tables itself calls it explicitly. */
int sdpgpu_finalize(sdpgpu_handle* h);
/* sdpgpu_finalize, then block until everything queued on the handle's stream has finished. */
int sdpgpu_synchronize(sdpgpu_handle* h);

/* ---- multi-GPU: the state axis sharded over the GPUs of one node ------------------------------------------
 * The reference's contract is "one call solves everything" (`getExpectedValue(initialState)`,
 * Recursion.java:89); with the state axis cut into world_size slabs that call becomes a sweep of per-slab period
 * kernels with ONE all-gather of V_t between periods (RCCL over xGMI, in place in the row the next period reads,
 * issued by this library -- no Python or torch in the data path).  Two ways to drive it:
 *
 *  (1) one process (or thread) per GPU:  every rank creates its handle with desc.rank / desc.world_size /
 *      desc.device, ONE rank calls sdpgpu_comm_unique_id and hands the 128 bytes to the others by whatever channel
 *      the host has (a file, a socket, MPI, torch.distributed's store), every rank calls sdpgpu_comm_prepare (local:
 *      device tables, RCCL load) and the ranks agree that all succeeded, every rank calls sdpgpu_comm_init (collective:
 *      ncclCommInitRank), then sdpgpu_solve_sharded.
 *  (2) one process that owns all the GPUs (a JVM calling through JNI): one handle per device, all in this process,
 *      and ONE call sdpgpu_solve_multi(handles, n, ...) -- the library builds the communicators itself
 *      (ncclCommInitAll), drives every device from the calling thread and groups the per-period all-gathers
 *      (default), or starts one host thread per rank (SDPGPU_SHARDED_THREADS: for slabs whose periods take tens of
 *      microseconds, where the launches of eight devices issued from one thread would queue up on the host).
 *      Handles that SHARE a device (a rehearsal of N ranks on one GPU) exchange their slabs by device-to-device
 *      copies instead, since RCCL refuses two ranks on one device; the slab arithmetic and results are the same.
 *
 * RCCL is loaded on first use (dlopen of librccl.so.1): a single-GPU caller never pays for it.  RCCL failures come
 * back as SDPGPU_ERR_DEVICE with ncclGetErrorString's text.  Policy tables stay sharded (sdpgpu_policy reads this
 * rank's slab); after the sweep every rank holds every full V_t with t >= 2, and V_1 complete only on its own slab
 * unless `gather_first` is set. */
#define SDPGPU_UNIQUE_ID_BYTES 128
int sdpgpu_comm_unique_id(void* out_id /* SDPGPU_UNIQUE_ID_BYTES */);
/* ABI 5.  The part of sdpgpu_comm_init that can fail on one rank ALONE (device tables do not fit, no device, RCCL does
 * not load), without entering a collective.  Call it on every rank, let the ranks agree over the host's channel that
 * all returned SDPGPU_OK, and only then call sdpgpu_comm_init: a rank that fails here has not left its peers blocked
 * inside ncclCommInitRank.  (sdpgpu_comm_init does the same work itself when this call was skipped.) */
int sdpgpu_comm_prepare(sdpgpu_handle* h);
/* Collective over the `world` ranks; rank / world must equal the handle's desc.rank / desc.world_size.  world = 1 is
 * allowed (a one-rank communicator: the collective path with nobody to talk to -- used by the tests). */
int sdpgpu_comm_init(sdpgpu_handle* h, const void* unique_id, int32_t rank, int32_t world);
int sdpgpu_comm_destroy(sdpgpu_handle* h);
/* The all-gather of the row of `period` alone (sdpgpu_exchange_ptr), enqueued on the handle's stream behind the
 * period's kernel -- for a caller that steps the periods itself with sdpgpu_run_period. */
int sdpgpu_exchange(sdpgpu_handle* h, int32_t period);
/* Whole backward sweep t = T..1 of this rank's slab with the exchanges in between; every rank calls it.
 * flags: SDPGPU_SHARDED_SYNC        wait for the stream before returning (else asynchronous, like sdpgpu_solve)
 *        SDPGPU_SHARDED_OVERLAP     run the all-gather of V_{t+1} on a second stream beside the INTERIOR part of
 *                                   period t (families with a bounded footprint; otherwise same as blocking)
 *        SDPGPU_SHARDED_GATHER_FIRST  also all-gather V_1 (needed only if every rank wants the whole V_1) */
#define SDPGPU_SHARDED_SYNC 1
#define SDPGPU_SHARDED_OVERLAP 2
#define SDPGPU_SHARDED_GATHER_FIRST 4
#define SDPGPU_SHARDED_THREADS 8 /* ABI 5, sdpgpu_solve_multi only: one host thread per rank (see there) */
int sdpgpu_solve_sharded(sdpgpu_handle* h, int32_t flags);
/* Way (2): handles[r] is the handle of rank r (desc.rank = r, desc.world_size = n, any devices; all n describing the
 * same problem -- family, grids, pmf sizes -- else SDPGPU_ERR_ARG).  The communicators are created at the first call
 * and kept in the handles.  Errors are reported on handles[0].  The calling thread's current device is restored. */
int sdpgpu_solve_multi(sdpgpu_handle** handles, int32_t n, int32_t flags);

/* ---- results ----------------------------------------------------------------------------- */
/* Copy V_period[0..n) to host (n <= num_states). */
int sdpgpu_values(sdpgpu_handle* h, int32_t period, double* out, int64_t n);
/* Copy this rank's slab of the arg-opt action INDEX table (action = index * step). */
int sdpgpu_policy(sdpgpu_handle* h, int32_t period, int32_t* out, int64_t lo, int64_t n);
/* Evaluate arbitrary states of period t against V_{t+1}: out_value/out_action_index get n entries. */
int sdpgpu_eval_states(sdpgpu_handle* h, int32_t period, int64_t n, const double* x, const double* cash,
                       const double* preq, double* out_value, int32_t* out_action_index);
/* The same with the second pipeline quantity per state (lead_time 2; preq2 may be NULL = zeros). */
int sdpgpu_eval_states2(sdpgpu_handle* h, int32_t period, int64_t n, const double* x, const double* cash,
                        const double* preq, const double* preq2, double* out_value, int32_t* out_action_index);
/* Reachable-set mask of period t (1 byte per state), forward-propagated from the ini_* state over
 * all feasible actions and all demands -- the key set the memoised recursion would build. */
int sdpgpu_reachable(sdpgpu_handle* h, int32_t period, uint8_t* out, int64_t n);

/* Roll the computed policy forward along n demand paths: the inner loops of
 * Simulation.simulateSDPGivenSamplNum (Simulation.java:59-69) and CashSimulation (CashSimulation.java:101-112)
 * as a batched table-lookup rollout, one path per lane.  demand[i*T + t] is the realised demand of path i in
 * period t+1, already rounded by the caller as the reference does (Math.round, Simulation.java:64);
 * discount[t] multiplies the immediate value of period t+1 (Math.pow(discountFactor, t) in CashSimulation,
 * all 1.0 for the undiscounted classes).  out_sum[i] = sum_t discount[t] * imm_t; out_valid[i] = 0 when
 * the path left the grid (possible only for unclamped families with demands outside the PMF support).
 * The start state is (1, ini_x, ini_cash, ini_preq[, desc.ini_preq2 with lead_time 2]); world_size must be 1.
 * Family SURVIVAL rolls RiskSimulation.simulateLostSale instead (RiskSimulation.java:213-234): out_sum[i] = 1 when
 * path i held negative cash at some point (else 0), bit 1 of out_valid[i] is set when a demand was lost on it, and
 * the order is forced to 0 in a state with negative cash; `discount` is ignored. */
int sdpgpu_simulate(sdpgpu_handle* h, int64_t n_paths, const double* demand, const double* discount, double ini_x,
                    double ini_cash, double ini_preq, double* out_sum, uint8_t* out_valid);

int sdpgpu_stats_get(sdpgpu_handle* h, sdpgpu_stats* out);

/* ABI 5: the launch plan of one period of THIS rank's slab, as the launcher would choose it now (descriptor, pmfs,
 * world size and the SDPGPU_WIN_* overrides read at create time).  Host arithmetic only: no device is touched, so a
 * caller (or a test without a GPU) can see what a period will cost in LDS before it runs.  Reported for the window
 * kernels of the backorder family (F1); other kernels return kernel = SDPGPU_KERNEL_GATHER and zeros.  A plan that
 * cannot run -- a forced register block that does not exist, a forced chunking that exceeds the 160 KiB of LDS of a
 * compute unit, chunk rows where ping-pong tables forbid them -- returns SDPGPU_ERR_ARG with the reason in
 * sdpgpu_last_error, which is also what sdpgpu_run_period then returns (before anything is launched). */
typedef struct sdpgpu_plan {
  int32_t kernel;            /* SDPGPU_KERNEL_WINDOW or SDPGPU_KERNEL_GATHER */
  int32_t r, s;              /* register block: actions x adjacent states per lane */
  int32_t chunks;            /* tasks per state tile (1: no chunk rows, no key atomics, no finalize pass) */
  int32_t chunk_blocks;      /* register blocks of the action axis per task */
  int32_t tiles, tasks;      /* state tiles of 64 s states of this slab; tiles x chunks */
  int32_t workgroups_per_cu; /* workgroups (four tasks each) the LDS lets a compute unit hold at once */
  int64_t lds_bytes;         /* dynamic LDS per workgroup */
} sdpgpu_plan;
int sdpgpu_plan_period(const sdpgpu_handle* h, int32_t period, sdpgpu_plan* out);

/* ---- reachable-set engine for the two-product lead-time family -------------------------------------
 * Replaces `new CashRecursionMultiLead(...).getExpectedValue(iniState)` / getAction
 * (src/sdp/cash/multiItem/CashRecursionMultiLead.java:31-95) for the lambdas of
 * src/cash/overdraft/MultiProductLeadtime.java:150-223 with DiscreteDistribution demands
 * (GetPmfMulti.java:157-172).  The state (I1, I2, preQ1, preQ2, cash) carries an un-rounded cash balance,
 * so there is no grid: the engine enumerates the reachable set level by level on the GPU, as the
 * reference's memoised recursion does on the host.  This is the family whose outputs the reference records
 * (MultiProductLeadtime.java:30-50). */
typedef struct sdpgpu_multilead {
  int32_t T;       /* horizon (TLength) */
  int32_t q_bound; /* actions (i, j), i, j in [0, Qbound) */
  double price[2], vari_cost[2], sal_value[2];
  double ini_cash, ini_i1, ini_i2;
  double r0, r1, r2, limit, interest_free;
  double min_inventory, max_inventory, min_cash, max_cash;
  double discount;
  double overhead[16]; /* overheadCost[t] */
  int32_t n1, n2;      /* demand points per product */
  double v1[16], p1[16], v2[16], p2[16];
  int32_t cash_int_cast; /* 1: `nextCash = (int) nextCash` (MultiProductLeadtime.java:219, commented out in the file as it
                            stands: "rounding states to save computing time") */
  int32_t reserved;
} sdpgpu_multilead;

/* final_value = iniCash + V_1(iniState) (MultiProductLeadtime.java:234), (q1, q2) = getAction(iniState);
 * states_per_period (T entries, may be NULL) = size of the reachable set per period, cells = (state, action,
 * demand) evaluations, gpu_ms = device time of expansion + recursion. */
int sdpgpu_multilead_solve(const sdpgpu_multilead* k, double* final_value, int32_t* q1, int32_t* q2,
                           int64_t* states_per_period, int64_t* cells, double* gpu_ms);
const char* sdpgpu_multilead_last_error(void);

/* Read-out of the whole memo of the two-product solvers -- what getCacheActions() / getOptTable() iterate
 * (CashRecursionMulti.java:140-168, CashRecursionMultiLead.java:92-101): register a table before a solve
 * (sdpgpu_multi_set_table, per thread; NULL clears it) and the next sdpgpu_multi*_solve of that thread fills it with one
 * row per visited state: period, the state tuple (q1 = q2 = 0 outside the lead-time family; `cash` holds R for the XR
 * family), V(state), and the chosen action pair (order-up-to levels for the XR family).  `rows` receives the number of
 * visited states; nothing is written when it exceeds `capacity` (call again with larger arrays).  Rows are grouped by
 * period, in no particular order inside a period. */
typedef struct sdpgpu_multi_table {
  int64_t capacity;
  int64_t rows;
  int32_t* period;
  double* i1;
  double* i2;
  double* q1;
  double* q2;
  double* cash;
  double* value;
  int32_t* a1;
  int32_t* a2;
} sdpgpu_multi_table;
void sdpgpu_multi_set_table(sdpgpu_multi_table* table);

/* sdp.cash.multiItem.CashRecursionMulti.getExpectedValue (CashRecursionMulti.java:82-116) over the lambdas of
 * cash.multiItem.MultiItemCash (MultiItemCash.java:66-118): two products, cash-limited orders
 * (variCost[0] * i + variCost[1] * j < cash + 0.1), no lead time, state (I1, I2, cash) truncated to ints by the
 * transition, the `> val + 0.1` scan.  Runs on the same reachable-set engine as sdpgpu_multilead_solve.
 * The joint pmf of period t+1 is the list GetPmfMulti.getPmf(t) returns (rows {d1, d2, probability}):
 * entries pmf_off[t] .. pmf_off[t+1]-1 of d1 / d2 / p. */
typedef struct sdpgpu_multicash {
  int32_t T;       /* horizon (TLength) */
  int32_t q_bound; /* actions (i, j), i, j in [0, Qbound) */
  double price[2], vari_cost[2], sal_price[2];
  double ini_cash, ini_i1, ini_i2;
  double min_inventory, max_inventory, min_cash, max_cash; /* min_cash >= 0 */
  double discount;
  const int32_t* pmf_off; /* T + 1 offsets */
  const double* d1;
  const double* d2;
  const double* p;
} sdpgpu_multicash;

/* final_value = iniCash + V_1(iniState) (MultiItemCash.java:132); cells counts the offered actions only (the
 * reference evaluates nothing else); errors through sdpgpu_multilead_last_error(). */
int sdpgpu_multicash_solve(const sdpgpu_multicash* k, double* final_value, int32_t* q1, int32_t* q2,
                           int64_t* states_per_period, int64_t* cells, double* gpu_ms);

/* sdp.cash.multiItem.CashRecursionMultiXR.getExpectedValue (CashRecursionMultiXR.java:60-96) over the lambdas of
 * cash.multiItem.MultiItemCashXR (MultiItemCashXR.java:92-148): state (x1, x2, R) with R = cash + variCost . x, actions
 * = order-up-to levels (y1, y2) in [(int) x, (int) x + Qbound), no cash limit on them, demands kept as doubles.  Same
 * descriptor as sdpgpu_multicash_solve (ini_cash is the R of the period-1 state, MultiItemCashXR.java:158) plus
 * depositeRate; final_value = iniCash + V_1(iniState) (:160), (y1, y2) = getAction(iniState). */
int sdpgpu_multixr_solve(const sdpgpu_multicash* k, double deposit_rate, double* final_value, int32_t* y1, int32_t* y2,
                         int64_t* states_per_period, int64_t* cells, double* gpu_ms);
/* Kernel time of period t of the last solve (ms), needs sdpgpu_set_profiling(h, 1). */
double sdpgpu_period_ms(sdpgpu_handle* h, int32_t period);
/* ABI 6: (state, action, demand) cells of period t on this rank's slab, as sdpgpu_stats.cells_evaluated sums them over the periods
 * run (bench.py prices a kernel's counters against the cells of ITS launches: period T of a family may offer fewer orders).  -1:
 * not counted yet (the period has not run) or not known per period. */
int64_t sdpgpu_period_cells(sdpgpu_handle* h, int32_t period);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* SDPGPU_H */
