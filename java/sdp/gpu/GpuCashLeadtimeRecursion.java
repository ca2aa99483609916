package sdp.gpu;

import java.util.ArrayList;
import java.util.function.Function;

import sdp.cash.CashLeadtimeState;
import sdp.inventory.ImmediateValue.ImmediateValueFunction;
import sdp.inventory.StateTransition.StateTransitionFunction;

/**
 * Drop-in for sdp.cash.CashLeadtimeRecursion (CashLeadtimeRecursion.java:28-46, :48-79, :97-106) on the MI355X engine:
 * same constructor arguments plus a functor descriptor, same public methods; MAX only, as the reference (:53, :70).
 * SOURCE ONLY (no JDK in the authoring image); the tested mirrors are stochastic-inventory_amd/recursion.py
 * (CashLeadtimeRecursion) and include/sdpgpu_mirror.hpp (sdp::gpu::CashLeadtimeRecursion).
 *
 * One deliberate difference: the reference's key comparator compares a cash value with itself (:37-41), which can alias
 * states that differ only in cash; the engine indexes by the full tuple (period, inventory, cash, preQ).
 *
 *   CashLeadtimeRecursion recursion = new CashLeadtimeRecursion(pmf, getFeasibleAction, stateTransition, immediateValue);
 *   GpuCashLeadtimeRecursion recursion = new GpuCashLeadtimeRecursion(pmf, getFeasibleAction, stateTransition,
 *           immediateValue, GpuCashLeadtimeRecursion.singleProductLeadtime(price, variCost, salvageValue, maxOrderQuantity,
 *                   minInventoryState, maxInventoryState, minCashState, maxCashState, r0, r2, r3, limit, interestFreeAmount,
 *                   overheadCosts));
 */
public class GpuCashLeadtimeRecursion {
	private final double[][][] pmf;
	private final StateTransitionFunction<CashLeadtimeState, Double, Double, CashLeadtimeState> stateTransition;
	private final ImmediateValueFunction<CashLeadtimeState, Double, Double, Double> immediateValue;
	private final Functor functor;
	private long handle;
	private boolean solved;
	private final double[][] values;
	private final int[][] policy;

	/** The scalar block plus the per-period overhead array of SingleProductLeadtime.java:35-36. */
	public static final class Functor {
		final GpuRecursion.Functor scalars;
		final double[] overheadCosts;

		Functor(GpuRecursion.Functor scalars, double[] overheadCosts) {
			this.scalars = scalars;
			this.overheadCosts = overheadCosts.clone();
		}
	}

	/**
	 * F5: the lambdas of SingleProductLeadtime.java:72-119 -- piecewise overdraft interest (:86-95), cash in hundredths
	 * (`Math.round(nextCash * 100) / 100.0`, :117), no order in the last period (:74-75).
	 */
	public static Functor singleProductLeadtime(double price, double variCost, double salvageValue,
			double maxOrderQuantity, double minInventoryState, double maxInventoryState, double minCashState,
			double maxCashState, double r0, double r2, double r3, double limit, double interestFreeAmount,
			double[] overheadCosts) {
		GpuRecursion.Functor f = GpuCashRecursion.cashConstraint(price, 0, variCost, 0, 0, overheadCosts[0], 0,
				salvageValue, 0, maxOrderQuantity, minInventoryState, maxInventoryState, minCashState, maxCashState, 100,
				100.0, false, 0);
		f.ints[0] = SdpGpu.FAMILY_CASH_LEADTIME;
		f.ints[4] = 1; // zero order in the last period
		f.doubles[21] = r0;
		f.doubles[22] = r2;
		f.doubles[23] = r3;
		f.doubles[24] = limit;
		f.doubles[25] = interestFreeAmount;
		return new Functor(f, overheadCosts);
	}

	public GpuCashLeadtimeRecursion(double[][][] pmf, Function<CashLeadtimeState, double[]> getFeasibleAction,
			StateTransitionFunction<CashLeadtimeState, Double, Double, CashLeadtimeState> stateTransition,
			ImmediateValueFunction<CashLeadtimeState, Double, Double, Double> immediateValue, Functor functor) {
		this.pmf = pmf;
		this.stateTransition = stateTransition;
		this.immediateValue = immediateValue;
		this.functor = functor;
		functor.scalars.ints[1] = SdpGpu.MAX;
		functor.scalars.ints[2] = pmf.length;
		this.handle = SdpGpu.create(functor.scalars.ints, functor.scalars.doubles);
		for (int t = 0; t < pmf.length; t++) {
			double[] d = new double[pmf[t].length], p = new double[pmf[t].length];
			for (int j = 0; j < d.length; j++) {
				d[j] = pmf[t][j][0];
				p[j] = pmf[t][j][1];
			}
			SdpGpu.setPmf(handle, t, d, p);
			SdpGpu.setOverhead(handle, t, functor.overheadCosts[t]);
		}
		this.values = new double[pmf.length][];
		this.policy = new int[pmf.length][];
	}

	public StateTransitionFunction<CashLeadtimeState, Double, Double, CashLeadtimeState> getStateTransitionFunction() {
		return stateTransition;
	}

	public ImmediateValueFunction<CashLeadtimeState, Double, Double, Double> getImmediateValueFunction() {
		return immediateValue;
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

	private double[] lookup(CashLeadtimeState s) {
		table(s.getPeriod());
		long idx = SdpGpu.stateIndex(handle, s.getPeriod(), s.getIniInventory(), s.getIniCash(), s.getPreQ());
		if (idx >= 0)
			return new double[] { values[s.getPeriod() - 1][(int) idx], policy[s.getPeriod() - 1][(int) idx] };
		double[] v = new double[1];
		int[] a = new int[1];
		SdpGpu.evalStates(handle, s.getPeriod(), new double[] { s.getIniInventory() }, new double[] { s.getIniCash() },
				new double[] { s.getPreQ() }, v, a);
		return new double[] { v[0], a[0] };
	}

	public double getExpectedValue(CashLeadtimeState state) {
		return lookup(state)[0];
	}

	public double getAction(CashLeadtimeState state) {
		return lookup(state)[1] * functor.scalars.doubles[0];
	}

	/** Rows {period, inventory, cash, preQ, Q} of the reachable states (CashLeadtimeRecursion.java:97-106). */
	public double[][] getOptTable() {
		ArrayList<double[]> rows = new ArrayList<>();
		double step = functor.scalars.doubles[0];
		for (int period = 1; period <= pmf.length; period++) {
			table(period);
			byte[] mask = new byte[values[period - 1].length];
			SdpGpu.reachable(handle, period, mask);
			double[] g = SdpGpu.grid(handle, period); // {x_lo, nx, nc, nq}; flat index = (iq * nx + ix) * nc + ic
			int nx = (int) g[1], nc = (int) g[2], nq = (int) g[3];
			for (int ix = 0; ix < nx; ix++)
				for (int iq = 0; iq < nq; iq++)
					for (int ic = 0; ic < nc; ic++) {
						int i = (iq * nx + ix) * nc + ic;
						if (mask[i] != 0)
							rows.add(new double[] { period, g[0] + ix * step, SdpGpu.cashValue(handle, ic), iq * step,
									policy[period - 1][i] * step });
					}
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
