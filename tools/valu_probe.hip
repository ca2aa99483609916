// valu_probe.hip -- measures what bounds the window kernel: fp64 add/mul issue rate per SIMD and the
// shader clock the chip holds under that load.  Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int CHAINS>
__global__ void fp64_chain(double* out, int iters, double p, double c, unsigned long long* clk) {
  double acc[CHAINS];
  double w = out[threadIdx.x & 63];
#pragma unroll
  for (int r = 0; r < CHAINS; ++r) acc[r] = (double)r;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int r = 0; r < CHAINS; ++r) {  // the window kernel's 5 ops per cell
      double imm = c + w;
      acc[r] += p * imm;
      acc[r] += p * w;
      w += 1.0;  // keep the compiler from hoisting
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
#pragma unroll
  for (int r = 0; r < CHAINS; ++r) s += acc[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) {
    clk[2 * blockIdx.x] = t1 - t0;
    clk[2 * blockIdx.x + 1] = r1 - r0;
  }
}

int main() {
  const int iters = 20000;
  double* out;
  unsigned long long* clk;
  hipMalloc(&out, 256 * 8192 * sizeof(double));
  hipMemset(out, 0, 256 * 8192 * sizeof(double));
  hipMalloc(&clk, 2 * 8192 * sizeof(unsigned long long));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int wg_per_cu : {1, 2, 4, 8}) {
    int blocks = 256 * wg_per_cu;  // 256 threads = 4 waves = one per SIMD per workgroup
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(fp64_chain<8>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.37, 1.5, clk);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
    }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * blocks);
    hipMemcpy(h.data(), clk, h.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0, real = 0;
    for (int b = 0; b < blocks; ++b) { cyc += h[2 * b]; real += h[2 * b + 1]; }
    cyc /= blocks; real /= blocks;
    double ghz = cyc / (real * 10.0);  // memrealtime ticks at 100 MHz
    double wave_instr = (double)iters * 8 * 6;  // 6 fp64 ops per chain step (5 + the w increment)
    double waves_per_simd = wg_per_cu;
    printf("waves/SIMD %d: %.3f ms, in-kernel clock %.3f GHz, cycles/wave-instr (per SIMD) %.2f, fp64 ops/s %.3e\n", wg_per_cu, ms,
           ghz, cyc / (wave_instr * waves_per_simd) * waves_per_simd, wave_instr * 64 * 4 * blocks / (ms * 1e-3));
  }
  return 0;
}
