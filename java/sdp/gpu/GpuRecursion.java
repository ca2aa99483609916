package sdp.gpu;

import java.util.Map;
import java.util.TreeMap;
import java.util.function.Function;

import sdp.inventory.ImmediateValue.ImmediateValueFunction;
import sdp.inventory.State;
import sdp.inventory.StateTransition.StateTransitionFunction;

/**
 * Drop-in for sdp.inventory.Recursion backed by the MI355X engine: same constructor arguments plus
 * a functor descriptor, same public methods.  A driver changes one line:
 *
 *   // Recursion recursion = new Recursion(OptDirection.MIN, pmf, getFeasibleAction, stateTransition, immediateValue);
 *   GpuRecursion recursion = new GpuRecursion(GpuRecursion.OptDirection.MIN, pmf, getFeasibleAction,
 *           stateTransition, immediateValue,
 *           Functor.backorder(fixedOrderingCost, variOrderingCost, holdingCost, penaltyCost,
 *                             minInventory, maxInventory, maxOrderQuantity, stepSize));
 *
 * The three lambdas are kept for the simulators (getStateTransitionFunction /
 * getImmediateValueFunction) -- the GPU evaluates the functor family.  SOURCE ONLY here (no JDK in
 * the authoring image); the tested mirror of this class is
 * stochastic-inventory_amd/recursion.py, which binds the same C entry points.
 */
public class GpuRecursion {

	public enum OptDirection {
		MIN, MAX
	}

	/** Scalar parameters of one lambda family, in the order of struct sdpgpu_desc. */
	public static final class Functor {
		final int[] ints = new int[12];
		final double[] doubles = new double[28];

		/** F1: the lambdas of CLSP / CLSPTesting / CLSPforDraw / LevelFitsS. */
		public static Functor backorder(double fixedOrderingCost, double variOrderingCost, double holdingCost,
				double penaltyCost, double minInventory, double maxInventory, double maxOrderQuantity,
				double stepSize) {
			Functor f = new Functor();
			f.ints[0] = SdpGpu.FAMILY_BACKORDER;
			f.ints[3] = 1; // clamp
			f.ints[7] = 0; // kernel auto
			f.ints[8] = -1; // current device
			f.ints[10] = 1; // world size
			f.ints[11] = 1; // store all values
			f.doubles[0] = stepSize;
			f.doubles[1] = minInventory;
			f.doubles[2] = maxInventory;
			f.doubles[3] = maxOrderQuantity;
			f.doubles[7] = fixedOrderingCost;
			f.doubles[8] = variOrderingCost;
			f.doubles[9] = holdingCost;
			f.doubles[10] = penaltyCost;
			f.doubles[16] = 1.0; // discount
			f.doubles[19] = 10;
			f.doubles[20] = 10;
			return f;
		}
	}

	private final double[][][] pmf;
	private final Function<State, double[]> getFeasibleActions;
	private final StateTransitionFunction<State, Double, Double, State> stateTransition;
	private final ImmediateValueFunction<State, Double, Double, Double> immediateValue;
	private final Functor functor;
	private long handle;
	private long[] rankHandles; // nGpus > 1: one handle per device, rank r on device r (handle == rankHandles[0])
	private boolean solved;
	private final double[][] values;
	private final int[][] policy;

	public GpuRecursion(OptDirection optDirection, double[][][] pmf, Function<State, double[]> getFeasibleAction,
			StateTransitionFunction<State, Double, Double, State> stateTransition,
			ImmediateValueFunction<State, Double, Double, Double> immediateValue, Functor functor) {
		this(optDirection, pmf, getFeasibleAction, stateTransition, immediateValue, functor, 1);
	}

	/**
	 * The same on nGpus devices of this node: the state axis is cut into nGpus slabs, getExpectedValue's first call
	 * is still ONE native call (SdpGpu.solveMulti -> sdpgpu_solve_multi: per-slab kernels + RCCL all-gather of V_t
	 * between periods).
	 */
	public GpuRecursion(OptDirection optDirection, double[][][] pmf, Function<State, double[]> getFeasibleAction,
			StateTransitionFunction<State, Double, Double, State> stateTransition,
			ImmediateValueFunction<State, Double, Double, Double> immediateValue, Functor functor, int nGpus) {
		this.pmf = pmf;
		this.getFeasibleActions = getFeasibleAction;
		this.stateTransition = stateTransition;
		this.immediateValue = immediateValue;
		this.functor = functor;
		functor.ints[1] = optDirection == OptDirection.MIN ? SdpGpu.MIN : SdpGpu.MAX;
		functor.ints[2] = pmf.length;
		this.rankHandles = new long[Math.max(1, nGpus)];
		try {
			for (int r = 0; r < rankHandles.length; r++) {
				int[] ints = functor.ints.clone();
				if (rankHandles.length > 1) {
					ints[8] = r; // device
					ints[9] = r; // rank
					ints[10] = rankHandles.length; // world size
				}
				rankHandles[r] = SdpGpu.create(ints, functor.doubles);
				for (int t = 0; t < pmf.length; t++) {
					double[] d = new double[pmf[t].length], p = new double[pmf[t].length];
					for (int j = 0; j < d.length; j++) {
						d[j] = pmf[t][j][0];
						p[j] = pmf[t][j][1];
					}
					SdpGpu.setPmf(rankHandles[r], t, d, p);
				}
			}
		} catch (RuntimeException e) {
			// a rank that cannot be created (or takes no pmf) must not leak the native handles of the ranks before it
			for (long h : rankHandles)
				if (h != 0)
					SdpGpu.destroy(h);
			throw e;
		}
		this.handle = rankHandles[0];
		this.values = new double[pmf.length][];
		this.policy = new int[pmf.length][];
	}

	public StateTransitionFunction<State, Double, Double, State> getStateTransitionFunction() {
		return stateTransition;
	}

	public ImmediateValueFunction<State, Double, Double, Double> getImmediateValueFunction() {
		return immediateValue;
	}

	/** Recursion.setTreeMapCacheAction: the dense tables are already in comparator order. */
	public void setTreeMapCacheAction() {
	}

	private void table(int period) {
		if (!solved) {
			if (rankHandles.length > 1)
				SdpGpu.solveMulti(rankHandles, true); // every V_t complete on rank 0, the policy stays sharded
			else
				SdpGpu.solve(handle);
			solved = true;
		}
		if (values[period - 1] == null) {
			int n = (int) SdpGpu.numStates(handle, period);
			values[period - 1] = new double[n];
			policy[period - 1] = new int[n];
			SdpGpu.values(handle, period, values[period - 1]);
			if (rankHandles.length == 1) {
				SdpGpu.policy(handle, period, policy[period - 1]);
			} else {
				for (long h : rankHandles) { // stitch the policy slabs together
					long[] s = SdpGpu.slab(h, period);
					int[] part = new int[(int) (s[2] - s[1])];
					SdpGpu.policySlab(h, period, s[1], part);
					System.arraycopy(part, 0, policy[period - 1], (int) s[1], part.length);
				}
			}
		}
	}

	public double getExpectedValue(State state) {
		table(state.getPeriod());
		long idx = SdpGpu.stateIndex(handle, state.getPeriod(), state.getIniInventory(), 0, 0);
		if (idx >= 0)
			return values[state.getPeriod() - 1][(int) idx];
		double[] v = new double[1];
		int[] a = new int[1];
		SdpGpu.evalStates(handle, state.getPeriod(), new double[] { state.getIniInventory() }, null, null, v, a);
		return v[0];
	}

	public double getAction(State state) {
		table(state.getPeriod());
		long idx = SdpGpu.stateIndex(handle, state.getPeriod(), state.getIniInventory(), 0, 0);
		if (idx >= 0)
			return policy[state.getPeriod() - 1][(int) idx] * functor.doubles[0];
		double[] v = new double[1];
		int[] a = new int[1];
		SdpGpu.evalStates(handle, state.getPeriod(), new double[] { state.getIniInventory() }, null, null, v, a);
		return a[0] * functor.doubles[0];
	}

	/** Rows {period, inventory, Q} of the reachable states, in (period, inventory) order. */
	public double[][] getOptTable() {
		java.util.ArrayList<double[]> rows = new java.util.ArrayList<>();
		for (int period = 1; period <= pmf.length; period++) {
			table(period);
			byte[] mask = new byte[values[period - 1].length];
			SdpGpu.reachable(handle, period, mask);
			double[] g = SdpGpu.grid(handle, period);
			for (int i = 0; i < mask.length; i++)
				if (mask[i] != 0)
					rows.add(new double[] { period, g[0] + i * functor.doubles[0],
							policy[period - 1][i] * functor.doubles[0] });
		}
		return rows.toArray(new double[0][]);
	}

	public Map<State, Double> getCacheActions() {
		Map<State, Double> m = new TreeMap<>((o1, o2) -> o1.getPeriod() != o2.getPeriod()
				? Integer.compare(o1.getPeriod(), o2.getPeriod())
				: Double.compare(o1.getIniInventory(), o2.getIniInventory()));
		for (double[] r : getOptTable())
			m.put(new State((int) r[0], r[1]), r[2]);
		return m;
	}

	public void close() {
		if (handle != 0) {
			for (long h : rankHandles)
				SdpGpu.destroy(h);
			handle = 0;
		}
	}
}
