package sdp.cash.multiItem;

import java.util.ArrayList;
import java.util.Map;
import java.util.TreeMap;
import java.util.function.Function;

import sdp.gpu.SdpGpu;
import sdp.inventory.ImmediateValue.ImmediateValueFunction;
import sdp.inventory.StateTransition.StateTransitionFunction;

/**
 * Drop-in for sdp.cash.multiItem.CashRecursionMultiXR (CashRecursionMultiXR.java:39-148) on the MI355X engine for the
 * lambdas of cash.multiItem.MultiItemCashXR.main (MultiItemCashXR.java:92-148): same constructor arguments plus the
 * scalars those lambdas close over.  SOURCE ONLY (no JDK in the authoring image); the tested mirror is
 * stochastic-inventory_amd/multiitem.py (CashRecursionMultiXR).  The sibling class for CashRecursionMulti /
 * MultiItemCash differs only in `model = 1` and Actions instead of double[] (sdpgpu_multicash_solve).
 *
 *   CashRecursionMultiXR recursion = new CashRecursionMultiXR(discountFactor, PmfMulti, buildActionList, stateTransition,
 *           immediateValue, T);                                                                            // before
 *   GpuCashRecursionMultiXR recursion = new GpuCashRecursionMultiXR(discountFactor, PmfMulti, buildActionList,
 *           stateTransition, immediateValue, T, price, variCost, salPrice, depositeRate, Qbound, minInventoryState,
 *           maxInventoryState, minCashState, maxCashState);                                               // after
 */
public class GpuCashRecursionMultiXR {
	private final double discountFactor;
	private final GetPmfMulti Pmf;
	private final int TLength, Qbound;
	private final double[] price, variCost, salPrice;
	private final double depositeRate, minInventoryState, maxInventoryState, minCashState, maxCashState;
	private final StateTransitionFunction<CashStateMultiXR, double[], double[], CashStateMultiXR> stateTransition;
	private final ImmediateValueFunction<CashStateMultiXR, double[], double[], Double> immediateValue;
	private final TreeMap<CashStateMultiXR, double[]> cacheActions;
	private final TreeMap<CashStateMultiXR, Double> cacheValues;

	public GpuCashRecursionMultiXR(double discountFactor, GetPmfMulti Pmf,
			Function<CashStateMultiXR, ArrayList<double[]>> buildActionList,
			StateTransitionFunction<CashStateMultiXR, double[], double[], CashStateMultiXR> stateTransition,
			ImmediateValueFunction<CashStateMultiXR, double[], double[], Double> immediateValue, int TLength,
			double[] price, double[] variCost, double[] salPrice, double depositeRate, int Qbound,
			double minInventoryState, double maxInventoryState, double minCashState, double maxCashState) {
		this.discountFactor = discountFactor;
		this.Pmf = Pmf;
		this.TLength = TLength;
		this.stateTransition = stateTransition;
		this.immediateValue = immediateValue;
		this.price = price;
		this.variCost = variCost;
		this.salPrice = salPrice;
		this.depositeRate = depositeRate;
		this.Qbound = Qbound;
		this.minInventoryState = minInventoryState;
		this.maxInventoryState = maxInventoryState;
		this.minCashState = minCashState;
		this.maxCashState = maxCashState;
		// the reference's key order (CashRecursionMultiXR.java:50-56)
		java.util.Comparator<CashStateMultiXR> keyComparator = (o1, o2) -> o1.getPeriod() != o2.getPeriod()
				? Integer.compare(o1.getPeriod(), o2.getPeriod())
				: o1.getIniInventory1() != o2.getIniInventory1() ? Double.compare(o1.getIniInventory1(), o2.getIniInventory1())
						: o1.getIniInventory2() != o2.getIniInventory2()
								? Double.compare(o1.getIniInventory2(), o2.getIniInventory2())
								: Double.compare(o1.getIniR(), o2.getIniR());
		this.cacheActions = new TreeMap<>(keyComparator);
		this.cacheValues = new TreeMap<>(keyComparator);
	}

	/** The first call solves from `initialState` (a period-1 state) and fills both maps with every visited state. */
	public double getExpectedValue(CashStateMultiXR initialState) {
		if (cacheValues.isEmpty()) {
			int[] off = new int[TLength + 1];
			ArrayList<double[]> all = new ArrayList<>();
			for (int t = 0; t < TLength; t++) {
				double[][] dAndP = Pmf.getPmf(t);
				off[t + 1] = off[t] + dAndP.length;
				for (double[] row : dAndP)
					all.add(row);
			}
			double[] d1 = new double[all.size()], d2 = new double[all.size()], p = new double[all.size()];
			for (int j = 0; j < d1.length; j++) {
				d1[j] = all.get(j)[0];
				d2[j] = all.get(j)[1];
				p[j] = all.get(j)[2];
			}
			double[] scalars = { price[0], price[1], variCost[0], variCost[1], salPrice[0], salPrice[1],
					initialState.getIniR(), initialState.getIniInventory1(), initialState.getIniInventory2(),
					minInventoryState, maxInventoryState, minCashState, maxCashState, discountFactor, depositeRate };
			for (double[] r : SdpGpu.multiSolve(2, TLength, Qbound, scalars, off, d1, d2, p)) {
				CashStateMultiXR s = new CashStateMultiXR((int) r[0], r[1], r[2], r[3]);
				cacheValues.put(s, r[4]);
				cacheActions.put(s, new double[] { r[5], r[6] });
			}
		}
		return cacheValues.get(initialState);
	}

	public double[] getAction(CashStateMultiXR state) {
		return cacheActions.get(state);
	}

	public Map<CashStateMultiXR, double[]> getCacheActions() {
		return cacheActions;
	}

	public StateTransitionFunction<CashStateMultiXR, double[], double[], CashStateMultiXR> getStateTransitionFunction() {
		return stateTransition;
	}

	public ImmediateValueFunction<CashStateMultiXR, double[], double[], Double> getImmediateValueFunction() {
		return immediateValue;
	}
}
