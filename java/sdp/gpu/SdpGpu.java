package sdp.gpu;

/**
 * Thin JNI binding of the C ABI in include/sdpgpu.h (one native method per entry point the Java
 * side needs).  SOURCE ONLY in this repository: the authoring image has no JDK, so this class and
 * java/jni/sdpgpu_jni.c are uncompiled and untested here; every parity test goes through the same
 * C ABI from Python.  A maintainer builds it with
 *
 *   javac -h java/jni -d build/classes java/sdp/gpu/*.java
 *   gcc -shared -fPIC -I$JAVA_HOME/include -I$JAVA_HOME/include/linux -Iinclude \
 *       -o libsdpgpu_jni.so java/jni/sdpgpu_jni.c -L stochastic-inventory_amd -lsdpgpu
 *
 * Handles are opaque longs; arrays cross as primitive arrays (GetPrimitiveArrayCritical in the
 * shim, never retained); errors become IllegalStateException carrying sdpgpu_last_error().
 */
public final class SdpGpu {
	static {
		System.loadLibrary("sdpgpu_jni");
	}

	private SdpGpu() {
	}

	/** Families: the closed-form lambda families of the in-scope drivers (sdpgpu_family). */
	public static final int FAMILY_BACKORDER = 1, FAMILY_LEADTIME = 2, FAMILY_CASH = 3, FAMILY_OVERDRAFT = 4,
			FAMILY_CASH_LEADTIME = 5, FAMILY_SURVIVAL = 6, FAMILY_STAFF = 7;
	public static final int MIN = 0, MAX = 1;

	/**
	 * sdpgpu_create: ints = {family, direction, periods, clampInventory, zeroOrderLastPeriod,
	 * cashRoundIntDiv, cashFormula, kernel, device, rank, worldSize, storeAllValues}; doubles in the
	 * order of struct sdpgpu_desc's double fields.
	 */
	public static native long create(int[] ints, double[] doubles);

	/**
	 * sdpgpu_create_custom: a driver whose lambdas are not one of the families hands them over as HIP device
	 * text (three functions, see include/sdpgpu.h) together with the constants they close over; `ints[0]` then
	 * only names the state shape and the loop.
	 */
	public static native long createCustom(int[] ints, double[] doubles, String functorSource, double[] params);

	public static native void destroy(long handle);

	/**
	 * sdpgpu_build_id (ABI 6): the 16-hex-digit digest of the sources libsdpgpu.so was built from. A JVM runs the classes it
	 * was given; the native half is a separate artefact -- log this next to the version of the jar.
	 */
	public static native String buildId();

	public static native void setPmf(long handle, int t, double[] demand, double[] prob);

	/**
	 * sdpgpu_set_level_pmf (STAFF family): prob[y * rowStride + j] = pmfs[t][y][j][1], rowLen[y] = pmfs[t][y].length.
	 */
	public static native void setLevelPmf(long handle, int t, double[] prob, int[] rowLen, int nRows, int rowStride);

	public static native void setOverhead(long handle, int t, double overheadCost);

	/**
	 * sdpgpu_getpmf: `new GetPmf(distributions, truncationQuantile, stepSize).getpmf()[t]` (variant 0, GetPmf.java:82-134)
	 * or CLSP.main's inline pmf (variant 1, CLSP.java:219-247) without SSJ: kinds[i] in {1 Poisson(a), 2 Normal(a, b),
	 * 3 UniformInt(a, b), 4 Gamma(shape a, rate b)}.  Returns rows {demand, probability}.
	 */
	public static native double[][] getPmf(int[] kinds, double[] a, double[] b, double truncationQuantile,
			double stepSize, int variant, int t);

	public static native void solve(long handle);

	/**
	 * sdpgpu_solve_multi: a JVM that owns every GPU of the node solves ONE problem on all of them with one call --
	 * handles[r] was created with ints[8] = device r, ints[9] = rank r, ints[10] = worldSize = handles.length.  The state
	 * axis is cut into handles.length slabs, the library creates the RCCL communicators itself (ncclCommInitAll) and
	 * all-gathers V_t between periods.  With gatherFirst every handle ends up holding the whole V_1 as well.
	 */
	public static native void solveMulti(long[] handles, boolean gatherFirst);

	/** One JVM per GPU instead: rank 0 draws the id (ncclGetUniqueId), the host ships the 128 bytes to the others. */
	public static native byte[] commUniqueId();

	/** sdpgpu_comm_init: collective over the ranks (ncclCommInitRank). */
	public static native void commInit(long handle, byte[] uniqueId, int rank, int worldSize);

	/** sdpgpu_solve_sharded: this rank's sweep with the per-period all-gathers; every rank calls it. */
	public static native void solveSharded(long handle, boolean overlap, boolean gatherFirst);

	/** {paddedRowLength, lo, hi}: this rank's slab of period t. */
	public static native long[] slab(long handle, int period);

	/** Policy indices of the slab [lo, lo + out.length) of this rank. */
	public static native void policySlab(long handle, int period, long lo, int[] out);

	public static native long numStates(long handle, int period);

	/** {x_lo, nx, nc, nq} of the period's grid. */
	public static native double[] grid(long handle, int period);

	public static native double cashValue(long handle, long ic);

	public static native long stateIndex(long handle, int period, double x, double cash, double preQ);

	public static native void values(long handle, int period, double[] out);

	public static native void policy(long handle, int period, int[] out);

	public static native void evalStates(long handle, int period, double[] x, double[] cash, double[] preQ,
			double[] outValue, int[] outActionIndex);

	public static native void reachable(long handle, int period, byte[] out);

	/**
	 * sdpgpu_multicash_solve (model 1, CashRecursionMulti over MultiItemCash's lambdas) / sdpgpu_multixr_solve (model 2,
	 * CashRecursionMultiXR over MultiItemCashXR's lambdas) with the memo read-out (sdpgpu_multi_set_table).
	 * scalars = {price0, price1, variCost0, variCost1, salPrice0, salPrice1, iniCashOrR, iniInventory1, iniInventory2,
	 * minInventory, maxInventory, minCash, maxCash, discountFactor, depositeRate}; pmfOff has T + 1 offsets into
	 * d1 / d2 / p (the rows of GetPmfMulti.getPmf(t)).  Returns rows {period, x1, x2, cashOrR, value, action1, action2}
	 * of every visited state, the period-1 state first.
	 */
	public static native double[][] multiSolve(int model, int T, int qBound, double[] scalars, int[] pmfOff, double[] d1,
			double[] d2, double[] p);
}
