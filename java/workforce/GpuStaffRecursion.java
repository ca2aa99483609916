package workforce;

import java.util.function.Function;

import sdp.gpu.SdpGpu;
import sdp.inventory.ImmediateValue.ImmediateValueFunction;
import sdp.inventory.StateTransition.StateTransitionFunction;

/**
 * Drop-in for workforce.StaffRecursion (StaffRecursion.java:42-118, :237-254) on the MI355X engine, in the
 * reference's own package because StaffState's fields are package-private (StaffState.java:5-6).
 * SOURCE ONLY (no JDK in the authoring image); the tested mirrors are stochastic-inventory_amd/workforce.py and
 * include/sdpgpu_mirror.hpp (sdp::gpu::StaffRecursion).
 *
 *   StaffRecursion recursion = new StaffRecursion(getFeasibleAction, stateTransition, immediateValue, pmf, T);    // before
 *   GpuStaffRecursion recursion = new GpuStaffRecursion(getFeasibleAction, stateTransition, immediateValue, pmf, T,
 *           GpuStaffRecursion.planning(fixCost, unitVariCost, salary, unitPenalty, minStaffNum, maxHireNum,
 *                   minX, maxX, iniStaffNum));                                                                     // after
 *   double opt = recursion.getExpectedValue(initialState);
 *
 * Built: getExpectedValue(StaffState), getAction, getOptTable -- what WorkforcePlanning.main (:104-112) and
 * WorkforceTesting.main (:116-124) call.  Not built: the G(y)-drawing variants (:127-330).
 */
public class GpuStaffRecursion implements AutoCloseable {
	/** The closed-form family the device evaluates: the lambdas of WorkforcePlanning.java:72-101. */
	public static final class Functor {
		final int[] ints = new int[12];
		final double[] doubles = new double[28];
		int[] minStaffNum;
	}

	/** clamped transition (WorkforcePlanning.java:84-89) */
	public static Functor planning(double fixCost, double unitVariCost, double salary, double unitPenalty,
			int[] minStaffNum, int maxHireNum, int minX, int maxX, int iniStaffNum) {
		Functor f = new Functor();
		f.ints[0] = SdpGpu.FAMILY_STAFF;
		f.ints[1] = SdpGpu.MIN;
		f.ints[3] = 1; // clamp
		f.ints[8] = -1; // current device
		f.ints[10] = 1; // world size
		f.ints[11] = 1; // store all values
		f.doubles[0] = 1; // step: heads
		f.doubles[1] = minX;
		f.doubles[2] = maxX;
		f.doubles[3] = maxHireNum;
		f.doubles[4] = iniStaffNum;
		f.doubles[7] = fixCost;
		f.doubles[8] = unitVariCost;
		f.doubles[9] = salary;
		f.doubles[10] = unitPenalty;
		f.doubles[16] = 1.0;
		f.doubles[19] = 10;
		f.doubles[20] = 10;
		f.minStaffNum = minStaffNum.clone();
		return f;
	}

	/** no clamp (WorkforceTesting.java:91-94): the staff range grows by maxHireNum a period from iniStaffNum */
	public static Functor testing(double fixCost, double unitVariCost, double salary, double unitPenalty,
			int[] minStaffNum, int maxHireNum, int iniStaffNum) {
		Functor f = planning(fixCost, unitVariCost, salary, unitPenalty, minStaffNum, maxHireNum, 0, 0, iniStaffNum);
		f.ints[3] = 0;
		return f;
	}

	private final long handle;
	private final int T;
	private final int iniStaffNum;
	private boolean solved = false;
	private final double[][] values;
	private final int[][] policy;
	private final Function<StaffState, int[]> getFeasibleAction;
	private final StateTransitionFunction<StaffState, Integer, Integer, StaffState> stateTransition;
	private final ImmediateValueFunction<StaffState, Integer, Integer, Double> immediateValue;

	public GpuStaffRecursion(Function<StaffState, int[]> getFeasibleAction,
			StateTransitionFunction<StaffState, Integer, Integer, StaffState> stateTransition,
			ImmediateValueFunction<StaffState, Integer, Integer, Double> immediateValue, double[][][][] pmf, int T,
			Functor functor) {
		this.getFeasibleAction = getFeasibleAction;
		this.stateTransition = stateTransition;
		this.immediateValue = immediateValue;
		this.T = T;
		this.iniStaffNum = (int) functor.doubles[4];
		functor.ints[2] = T;
		this.handle = SdpGpu.create(functor.ints, functor.doubles);
		for (int t = 0; t < T; t++) {
			int rows = pmf[t].length;
			int stride = 1;
			int[] rowLen = new int[rows];
			for (int y = 0; y < rows; y++) {
				rowLen[y] = pmf[t][y].length;
				stride = Math.max(stride, rowLen[y]);
			}
			double[] prob = new double[rows * stride];
			for (int y = 0; y < rows; y++)
				for (int j = 0; j < rowLen[y]; j++) {
					if (pmf[t][y][j][0] != j)
						throw new IllegalArgumentException("pmf[t][y][j][0] must be j");
					prob[y * stride + j] = pmf[t][y][j][1];
				}
			SdpGpu.setLevelPmf(handle, t, prob, rowLen, rows, stride);
			SdpGpu.setOverhead(handle, t, functor.minStaffNum[t]);
		}
		this.values = new double[T][];
		this.policy = new int[T][];
	}

	public StateTransitionFunction<StaffState, Integer, Integer, StaffState> getStateTransitionFunction() {
		return stateTransition;
	}

	public ImmediateValueFunction<StaffState, Integer, Integer, Double> getImmediateValueFunction() {
		return immediateValue;
	}

	private int indexOf(StaffState state) {
		if (!solved) {
			SdpGpu.solve(handle);
			solved = true;
		}
		int t = state.period - 1;
		if (values[t] == null) {
			int n = (int) SdpGpu.numStates(handle, state.period);
			values[t] = new double[n];
			policy[t] = new int[n];
			SdpGpu.values(handle, state.period, values[t]);
			SdpGpu.policy(handle, state.period, policy[t]);
		}
		long idx = SdpGpu.stateIndex(handle, state.period, state.iniStaffNum, 0, 0);
		if (idx < 0)
			throw new IllegalStateException(state + " lies outside the staff numbers the recursion can reach");
		return (int) idx;
	}

	public double getExpectedValue(StaffState state) {
		int i = indexOf(state);
		return values[state.period - 1][i];
	}

	public int getAction(StaffState state) {
		int i = indexOf(state);
		return policy[state.period - 1][i];
	}

	/** Rows {period, iniStaffNum, action} of the visited states in comparator order (StaffRecursion.java:245-254). */
	public double[][] getOptTable() {
		indexOf(new StaffState(1, iniStaffNum));
		java.util.ArrayList<double[]> rows = new java.util.ArrayList<>();
		for (int period = 1; period <= T; period++) {
			indexOf(new StaffState(period, (int) SdpGpu.grid(handle, period)[0]));
			byte[] mask = new byte[values[period - 1].length];
			SdpGpu.reachable(handle, period, mask);
			double xLo = SdpGpu.grid(handle, period)[0];
			for (int i = 0; i < mask.length; i++)
				if (mask[i] != 0)
					rows.add(new double[] { period, xLo + i, policy[period - 1][i] });
		}
		return rows.toArray(new double[0][]);
	}

	@Override
	public void close() {
		SdpGpu.destroy(handle);
	}
}
