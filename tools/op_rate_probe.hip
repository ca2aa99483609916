// op_rate_probe.hip -- issue cost (cycles per wave instruction per SIMD, 4 waves/SIMD) of the fp64 instructions the
// kernels lean on: add, mul, floor, rndne, cvt_i32_f64, cvt_f64_i32, min/max, cmp.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o tools/op_rate_probe tools/op_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

enum Op { ADD, MUL, FLOOR, RNDNE, CVT_I32, CVT_F64, MINMAX, CMPSEL, N_OPS };
static const char* kNames[N_OPS] = {"v_add_f64", "v_mul_f64", "v_floor_f64", "v_rndne_f64", "v_cvt_i32_f64", "v_cvt_f64_i32",
                                    "v_min/max_f64", "v_cmp_f64+cndmask"};

template <int OP>
__global__ void chain(double* out, int iters, unsigned long long* clk) {
  constexpr int C = 8;
  double a[C];
  int k[C];
#pragma unroll
  for (int r = 0; r < C; ++r) {
    a[r] = out[(threadIdx.x + r) & 63] + 1.25 * r;
    k[r] = r;
  }
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < C; ++r) {
      if (OP == ADD) a[r] = a[r] + 1.5;
      if (OP == MUL) a[r] = a[r] * 1.0000001;
      if (OP == FLOOR) asm volatile("v_floor_f64 %0, %1" : "=v"(a[r]) : "v"(a[r]));
      if (OP == RNDNE) asm volatile("v_rndne_f64 %0, %1" : "=v"(a[r]) : "v"(a[r]));
      if (OP == CVT_I32) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(k[r]) : "v"(a[r]));
      if (OP == CVT_F64) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(a[r]) : "v"(k[r]));
      if (OP == MINMAX) asm volatile("v_min_f64 %0, %1, %2" : "=v"(a[r]) : "v"(a[r]), "v"(a[(r + 1) % C]));
      if (OP == CMPSEL) a[r] = a[r] > 3.0 ? a[(r + 1) % C] : a[r];
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
#pragma unroll
  for (int r = 0; r < C; ++r) s += a[r] + k[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) clk[blockIdx.x] = t1 - t0;
}

template <int OP>
void run(double* out, unsigned long long* clk) {
  const int iters = 4000, blocks = 256 * 4;  // 4 workgroups of 4 waves per CU: 4 waves per SIMD
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(chain<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, clk);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks);
  hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
  double cyc = 0;
  for (auto v : h) cyc += (double)v;
  cyc /= blocks;
  // s_memtime counts at 100 MHz on this part: report relative to v_add_f64 instead of absolute cycles
  printf("%-22s %10.0f ticks for %d x 8 instructions x 4 waves/SIMD\n", kNames[OP], cyc, iters);
}

int main() {
  double* out;
  unsigned long long* clk;
  hipMalloc(&out, 256 * 4096 * sizeof(double));
  hipMemset(out, 0, 256 * 4096 * sizeof(double));
  hipMalloc(&clk, 8192 * sizeof(unsigned long long));
  run<ADD>(out, clk);
  run<MUL>(out, clk);
  run<FLOOR>(out, clk);
  run<RNDNE>(out, clk);
  run<CVT_I32>(out, clk);
  run<CVT_F64>(out, clk);
  run<MINMAX>(out, clk);
  run<CMPSEL>(out, clk);
  return 0;
}
