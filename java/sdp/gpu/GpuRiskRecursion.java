package sdp.gpu;

import java.util.function.Function;

import sdp.cash.CashState;
import sdp.cash.RiskState;
import sdp.inventory.ImmediateValue.ImmediateValueFunction;
import sdp.inventory.StateTransition.StateTransitionFunction;

/**
 * Drop-in for sdp.cash.RiskRecursion (RiskRecursion.java:31-46, :65-108, :123-132) on the MI355X engine:
 * the survival-probability recursion over the lambdas of cash.risk.cashSurvival (cashSurvival.java:98-143).
 * SOURCE ONLY (no JDK in the authoring image); the tested mirrors are stochastic-inventory_amd/recursion.py
 * (RiskRecursion) and include/sdpgpu_mirror.hpp (sdp::gpu::RiskRecursion).
 *
 *   RiskRecursion recursion = new RiskRecursion(pmf, getFeasibleAction, stateTransition, immediateValue);   // before
 *   GpuRiskRecursion recursion = new GpuRiskRecursion(pmf, getFeasibleAction, stateTransition, immediateValue,
 *           GpuRiskRecursion.survival(price[0], fixOrderCost, variCost[0], holdingCost, depositeRate,
 *                   overheadCosts[0], salvageValue, maxOrderQuantity, minInventoryState, maxInventoryState,
 *                   minCashState, maxCashState));                                                            // after
 *   double finalValue = recursion.getSurvProb(initialState);
 */
public class GpuRiskRecursion {
	private final GpuCashRecursion core;
	private final StateTransitionFunction<RiskState, Double, Double, RiskState> stateTransition;
	private final ImmediateValueFunction<RiskState, Double, Double, Double> immediateValue;

	/** F6: orders limited by cash / variCost, no end-cash penalty, Math.round(nextCash * 1) / 1. */
	public static GpuRecursion.Functor survival(double price, double fixOrderCost, double variCost, double holdingCost,
			double depositeRate, double overheadCost, double salvageValue, double maxOrderQuantity,
			double minInventoryState, double maxInventoryState, double minCashState, double maxCashState) {
		GpuRecursion.Functor f = GpuCashRecursion.cashConstraint(price, fixOrderCost, variCost, holdingCost, depositeRate,
				overheadCost, 0, salvageValue, 0, maxOrderQuantity, minInventoryState, maxInventoryState, minCashState,
				maxCashState, 1, 1, true, 0);
		f.ints[0] = SdpGpu.FAMILY_SURVIVAL;
		return f;
	}

	public GpuRiskRecursion(double[][][] pmf, Function<RiskState, double[]> getFeasibleAction,
			StateTransitionFunction<RiskState, Double, Double, RiskState> stateTransition,
			ImmediateValueFunction<RiskState, Double, Double, Double> immediateValue, GpuRecursion.Functor functor) {
		this.stateTransition = stateTransition;
		this.immediateValue = immediateValue;
		// the engine never calls the lambdas; the core only stores them for the simulators
		this.core = new GpuCashRecursion(GpuRecursion.OptDirection.MAX, pmf, null, null, null, 1.0, functor);
	}

	public StateTransitionFunction<RiskState, Double, Double, RiskState> getStateTransitionFunction() {
		return stateTransition;
	}

	public ImmediateValueFunction<RiskState, Double, Double, Double> getImmediateValueFunction() {
		return immediateValue;
	}

	public void setTreeMapCacheAction() {
	}

	public double getSurvProb(RiskState initialState) {
		return core.getExpectedValue((CashState) initialState);
	}

	public double getAction(RiskState state) {
		return core.getAction((CashState) state);
	}

	/** Rows {period, inventory, cash, bankruptBefore (always 0, RiskState.java:17), Q}. */
	public double[][] getOptTable() {
		double[][] t = core.getOptTable();
		double[][] out = new double[t.length][];
		for (int i = 0; i < t.length; i++)
			out[i] = new double[] { t[i][0], t[i][1], t[i][2], 0, t[i][3] };
		return out;
	}

	public void close() {
		core.close();
	}
}
