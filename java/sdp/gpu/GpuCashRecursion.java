package sdp.gpu;

import java.util.ArrayList;
import java.util.function.Function;

import sdp.cash.CashState;
import sdp.inventory.ImmediateValue.ImmediateValueFunction;
import sdp.inventory.StateTransition.StateTransitionFunction;

/**
 * Drop-in for sdp.cash.CashRecursion (CashRecursion.java:39-56, :79-140, :197-218) on the MI355X
 * engine: same constructor arguments plus a functor descriptor, same public methods.  SOURCE ONLY
 * (no JDK in the authoring image); the tested mirror is stochastic-inventory_amd/recursion.py.
 *
 *   CashRecursion recursion = new CashRecursion(OptDirection.MAX, pmf, getFeasibleAction, stateTransition,
 *                                               immediateValue, discountFactor);                 // before
 *   GpuCashRecursion recursion = new GpuCashRecursion(GpuRecursion.OptDirection.MAX, pmf, getFeasibleAction,
 *           stateTransition, immediateValue, discountFactor,
 *           GpuCashRecursion.cashConstraint(price, fixOrderCost, variCost, holdingCost, depositeRate, overheadCost,
 *                   overheadRate, salvageValue, penaltyCost, maxOrderQuantity, minInventoryState, maxInventoryState,
 *                   minCashState, maxCashState, 10, 10.0, false, 0));                               // after
 */
public class GpuCashRecursion {
	private final double[][][] pmf;
	private final StateTransitionFunction<CashState, Double, Double, CashState> stateTransition;
	private final ImmediateValueFunction<CashState, Double, Double, Double> immediateValue;
	private final GpuRecursion.Functor functor;
	private long handle;
	private boolean solved;
	private final double[][] values;
	private final int[][] policy;

	/** F3: the lambdas of CashConstraint.java:95-133 (formula 0) / CashConstraintTesting.java:110-148 (formula 1). */
	public static GpuRecursion.Functor cashConstraint(double price, double fixOrderCost, double variCost,
			double holdingCost, double depositeRate, double overheadCost, double overheadRate, double salvageValue,
			double penaltyCost, double maxOrderQuantity, double minInventoryState, double maxInventoryState,
			double minCashState, double maxCashState, double roundMult, double roundDiv, boolean longDivision,
			int formula) {
		GpuRecursion.Functor f = GpuRecursion.Functor.backorder(fixOrderCost, variCost, holdingCost, penaltyCost,
				minInventoryState, maxInventoryState, maxOrderQuantity, 1);
		f.ints[0] = SdpGpu.FAMILY_CASH;
		f.ints[5] = longDivision ? 1 : 0;
		f.ints[6] = formula;
		f.doubles[11] = price;
		f.doubles[12] = salvageValue;
		f.doubles[13] = depositeRate;
		f.doubles[14] = overheadCost;
		f.doubles[15] = overheadRate;
		f.doubles[17] = minCashState;
		f.doubles[18] = maxCashState;
		f.doubles[19] = roundMult;
		f.doubles[20] = roundDiv;
		return f;
	}

	public GpuCashRecursion(GpuRecursion.OptDirection optDirection, double[][][] pmf,
			Function<CashState, double[]> getFeasibleAction,
			StateTransitionFunction<CashState, Double, Double, CashState> stateTransition,
			ImmediateValueFunction<CashState, Double, Double, Double> immediateValue, double discountFactor,
			GpuRecursion.Functor functor) {
		this.pmf = pmf;
		this.stateTransition = stateTransition;
		this.immediateValue = immediateValue;
		this.functor = functor;
		functor.ints[1] = optDirection == GpuRecursion.OptDirection.MIN ? SdpGpu.MIN : SdpGpu.MAX;
		functor.ints[2] = pmf.length;
		functor.doubles[16] = discountFactor;
		this.handle = SdpGpu.create(functor.ints, functor.doubles);
		for (int t = 0; t < pmf.length; t++) {
			double[] d = new double[pmf[t].length], p = new double[pmf[t].length];
			for (int j = 0; j < d.length; j++) {
				d[j] = pmf[t][j][0];
				p[j] = pmf[t][j][1];
			}
			SdpGpu.setPmf(handle, t, d, p);
		}
		this.values = new double[pmf.length][];
		this.policy = new int[pmf.length][];
	}

	public StateTransitionFunction<CashState, Double, Double, CashState> getStateTransitionFunction() {
		return stateTransition;
	}

	public ImmediateValueFunction<CashState, Double, Double, Double> getImmediateValueFunction() {
		return immediateValue;
	}

	public void setTreeMapCacheAction() {
	}

	private void table(int period) {
		if (!solved) {
			SdpGpu.solve(handle);
			solved = true;
		}
		if (values[period - 1] == null) {
			int n = (int) SdpGpu.numStates(handle, period);
			values[period - 1] = new double[n];
			policy[period - 1] = new int[n];
			SdpGpu.values(handle, period, values[period - 1]);
			SdpGpu.policy(handle, period, policy[period - 1]);
		}
	}

	private double[] lookup(CashState s) {
		table(s.getPeriod());
		long idx = SdpGpu.stateIndex(handle, s.getPeriod(), s.getIniInventory(), s.getIniCash(), 0);
		if (idx >= 0)
			return new double[] { values[s.getPeriod() - 1][(int) idx], policy[s.getPeriod() - 1][(int) idx] };
		double[] v = new double[1];
		int[] a = new int[1];
		SdpGpu.evalStates(handle, s.getPeriod(), new double[] { s.getIniInventory() },
				new double[] { s.getIniCash() }, null, v, a);
		return new double[] { v[0], a[0] };
	}

	public double getExpectedValue(CashState state) {
		return lookup(state)[0];
	}

	public double getAction(CashState state) {
		return lookup(state)[1] * functor.doubles[0];
	}

	/** Rows {period, inventory, cash, Q} of the reachable states (CashRecursion.java:209-218). */
	public double[][] getOptTable() {
		ArrayList<double[]> rows = new ArrayList<>();
		for (int period = 1; period <= pmf.length; period++) {
			table(period);
			byte[] mask = new byte[values[period - 1].length];
			SdpGpu.reachable(handle, period, mask);
			double[] g = SdpGpu.grid(handle, period); // {x_lo, nx, nc, nq}
			int nc = (int) g[2];
			for (int i = 0; i < mask.length; i++)
				if (mask[i] != 0)
					rows.add(new double[] { period, g[0] + (i / nc) * functor.doubles[0],
							SdpGpu.cashValue(handle, i % nc), policy[period - 1][i] * functor.doubles[0] });
		}
		return rows.toArray(new double[0][]);
	}

	public void close() {
		if (handle != 0) {
			SdpGpu.destroy(handle);
			handle = 0;
		}
	}
}
