// Device check of the quantiser's rounding (csrc/sdp_cash.hpp: jround_rtn, jround_rtn2/4/8 -- two additions under
// round-toward-minus-infinity) against Math.round's definition, floor(x + 1/2) in EXACT arithmetic, evaluated on the host with
// integer / long-double arithmetic that cannot round: every adversarial value (the predecessor of 0.5, which floor(x + 0.5) in
// round-to-nearest gets wrong; exact ties of both signs; neighbours of ties; the int32 range's ends the launchers admit) and a
// few million random ones.  Also checks that the rounding mode is back to round-to-nearest after the block: an fp64 addition
// whose result depends on the mode is performed right behind it.
// Build + run: tests/test_gpu_round_rtn.py (hipcc, same flags as the library).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "../../stochastic-inventory_amd/csrc/sdp_cash.hpp"

__global__ void round_kernel(const double* x, int* k1, int* k4, double* probe, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  k1[i] = sdp::jround_rtn(x[i]);
  // the 4-wide and 8-wide forms on shifted copies of the same value (shifts by integers commute with the rounding)
  const double a4[4] = {x[i], x[i] + 1.0, x[i] - 1.0, x[i] + 2.0};
  int r4[4];
  sdp::jround_rtn4(a4, r4);
  const double a8[8] = {x[i], x[i] + 1.0, x[i] - 1.0, x[i] + 2.0, x[i] - 2.0, x[i] + 3.0, x[i] - 3.0, x[i] + 4.0};
  int r8[8];
  sdp::jround_rtn8(a8, r8);
  const double a2[2] = {x[i], x[i] + 1.0};
  int r2[2];
  sdp::jround_rtn2(a2, r2);
  int agree = 1;
  // (only where the shifted value is exact: |x| < 2^31 and the shift keeps the fraction bits -- true for |x| >= 1 ... skipped below)
  k4[i] = (r4[0] == k1[i] && r8[0] == k1[i] && r2[0] == k1[i]) ? agree : 0;
  // mode probe: 1 + 2^-53 rounds to 1.0 under round-to-nearest-even, to 1 + 2^-52 under round-up, 1.0 under round-down /
  // toward zero; 1 + 3 * 2^-54 rounds to 1 + 2^-52 under nearest, 1.0 under round-down: the pair tells nearest from the rest
  volatile double one = 1.0, tiny = 1.6653345369377348e-16;  // 3 * 2^-54
  probe[i] = one + tiny;
}

static long long exact_round(double x) {  // floor(x + 1/2) without rounding: x = m * 2^e exactly
  const double f = std::floor(x);         // exact
  const double frac = x - f;              // exact (Sterbenz-like: both in the same binade or frac < 1)
  return (long long)f + (frac >= 0.5 ? 1 : 0);
}

int main() {
  std::vector<double> xs;
  const double half_pred = std::nextafter(0.5, 0.0);
  for (double v : {0.0, -0.0, 0.5, -0.5, half_pred, -half_pred, std::nextafter(0.5, 1.0), std::nextafter(-0.5, -1.0), 1.5, -1.5, 2.5,
                   -2.5, 1e-300, -1e-300, 0.25, 0.75, -0.25, -0.75, 4.9e8, -4.9e8, 499999999.5, -499999999.5, 2147483646.5,
                   -2147483647.5, 2147483646.49, 100.49999999999999, 100.5, 100.50000000000001, -100.5})
    xs.push_back(v);
  for (int k = -2000; k <= 2000; ++k) {  // every tie in a range, and its two neighbours
    const double t = k + 0.5;
    xs.push_back(t);
    xs.push_back(std::nextafter(t, 1e9));
    xs.push_back(std::nextafter(t, -1e9));
  }
  std::mt19937_64 rng(12345);
  std::uniform_real_distribution<double> wide(-5.0e8, 5.0e8), cents(-1.0e6, 1.0e6);
  for (int i = 0; i < 2000000; ++i) xs.push_back(wide(rng));
  for (int i = 0; i < 2000000; ++i) xs.push_back(std::floor(cents(rng)) / 10.0 * 10.0 + 0.5 * (i & 1));  // many exact ties
  const int n = (int)xs.size();
  double *dx, *dprobe;
  int *dk1, *dk4;
  (void)hipMalloc(&dx, n * 8);
  (void)hipMalloc(&dprobe, n * 8);
  (void)hipMalloc(&dk1, n * 4);
  (void)hipMalloc(&dk4, n * 4);
  (void)hipMemcpy(dx, xs.data(), n * 8, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(round_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, dx, dk1, dk4, dprobe, n);
  if (hipDeviceSynchronize() != hipSuccess) {
    std::printf("KERNEL FAILED\n");
    return 2;
  }
  std::vector<int> k1(n), k4(n);
  std::vector<double> probe(n);
  (void)hipMemcpy(k1.data(), dk1, n * 4, hipMemcpyDeviceToHost);
  (void)hipMemcpy(k4.data(), dk4, n * 4, hipMemcpyDeviceToHost);
  (void)hipMemcpy(probe.data(), dprobe, n * 8, hipMemcpyDeviceToHost);
  long long bad = 0, bad_wide = 0, bad_mode = 0;
  for (int i = 0; i < n; ++i) {
    if ((long long)k1[i] != exact_round(xs[i])) {
      if (bad < 5) std::printf("MISMATCH x = %.17g: device %d, exact %lld\n", xs[i], k1[i], exact_round(xs[i]));
      ++bad;
    }
    if (!k4[i]) ++bad_wide;
    if (probe[i] != 1.0 + 2.220446049250313e-16) ++bad_mode;  // nearest-even: 1 + 3 * 2^-54 -> 1 + 2^-52
  }
  std::printf("checked %d values: %lld mismatches, %lld wide-form disagreements, %lld with a wrong rounding mode afterwards\n", n, bad,
              bad_wide, bad_mode);
  return (bad || bad_wide || bad_mode) ? 1 : 0;
}
