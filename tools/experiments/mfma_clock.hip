// What does a register-only v_mfma_f32_32x32x2_f32 loop sustain on THIS device, and at what shader clock?
// (124.8 TF measured in round 1 vs 155 TF in MI355X_MICROARCH.md.)
//
// One launch = 256 x WGS workgroups of 64*WPS*4 threads (WPS waves per SIMD), every wave issues
// `iters` x 16 back-to-back MFMAs on 4 independent accumulators, operands in registers, no memory
// traffic.  Wave 0 of every workgroup stamps s_memtime (shader cycles) and s_memrealtime (100 MHz)
// around its loop: clock = d(memtime) / d(memrealtime) x 100 MHz (median over workgroups).
// Bursts of ~0.5 / 5 / 50 ms, cold (after 1 s idle) and hot (after ~2 s of back-to-back launches).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_clock tools/experiments/mfma_clock.hip && /tmp/mfma_clock
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <unistd.h>

#include <algorithm>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void mfma_burst(float* sink, unsigned long long* stamps, int iters) {
  f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
  // "random" full-range operands (DVFS depends on the data: zeros read high)
  unsigned h = (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
  float x = (float)(int)(h & 0xffff) * (1.0f / 32768.0f) - 1.0f;
  float y = (float)(int)((h >> 16) & 0xffff) * (1.0f / 32768.0f) - 1.0f;
  unsigned long long t0 = 0, r0 = 0;
  const bool stamp = threadIdx.x == 0;
  if (stamp) {
    t0 = __builtin_amdgcn_s_memtime();
    r0 = __builtin_amdgcn_s_memrealtime();
  }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, x, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, x, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(y, y, a3, 0, 0, 0);
    }
  }
  float s = 0;
  for (int r = 0; r < 16; ++r) s += a0[r] + a1[r] + a2[r] + a3[r];
  if (stamp) {
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    stamps[2 * blockIdx.x] = t1 - t0;
    stamps[2 * blockIdx.x + 1] = r1 - r0;
  }
  if (s == 12345.678f) sink[threadIdx.x] = s;
}

int main() {
  float* sink;
  unsigned long long* stamps;
  const int max_wgs = 256 * 8;
  hipMalloc(&sink, 4096 * sizeof(float));
  hipMalloc(&stamps, 2 * max_wgs * sizeof(unsigned long long));
  std::vector<unsigned long long> host(2 * max_wgs);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  auto run = [&](int wps, int wgs_per_cu, int iters, const char* tag) {
    const int wgs = 256 * wgs_per_cu, threads = 256 * wps;
    hipMemset(stamps, 0, 2 * max_wgs * sizeof(unsigned long long));
    hipEventRecord(e0);
    hipLaunchKernelGGL(mfma_burst, dim3(wgs), dim3(threads), 0, 0, sink, stamps, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(host.data(), stamps, 2 * wgs * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    std::vector<double> clk;
    for (int i = 0; i < wgs; ++i)
      if (host[2 * i + 1]) clk.push_back((double)host[2 * i] / (double)host[2 * i + 1] * 0.1);   // GHz
    std::sort(clk.begin(), clk.end());
    const double flops = (double)wgs * wps * 4 * iters * 16 * 4096.0;
    printf("%-5s waves/SIMD %d  wg/CU %d  iters %7d : %8.3f ms  %6.1f TFLOP/s  clock median %.3f GHz (min %.3f max %.3f)\n",
           tag, wps * wgs_per_cu, wgs_per_cu, iters, ms, flops / ms * 1e-9, clk.empty() ? 0.0 : clk[clk.size() / 2],
           clk.empty() ? 0.0 : clk.front(), clk.empty() ? 0.0 : clk.back());
    return ms;
  };
  // warm the code object
  run(1, 1, 100, "warm");
  // one wave per SIMD x 16 MFMA x 64 cycles = 1024 cycles per iteration = 0.43 us at 2.4 GHz
  const int it_short = 1200, it_mid = 12000, it_long = 120000;
  for (int wps : {1, 2, 4}) {
    sleep(1);
    run(wps, 1, it_short / wps, "cold");
    sleep(1);
    run(wps, 1, it_mid / wps, "cold");
    sleep(1);
    run(wps, 1, it_long / wps, "cold");
    // hot: ~2 s of back-to-back launches, then the three bursts without a pause
    for (int k = 0; k < 40; ++k) hipLaunchKernelGGL(mfma_burst, dim3(256), dim3(256 * wps), 0, 0, sink, stamps, it_long / wps);
    hipDeviceSynchronize();
    run(wps, 1, it_short / wps, "hot");
    run(wps, 1, it_mid / wps, "hot");
    run(wps, 1, it_long / wps, "hot");
  }
  // the round-1 shape: 4 workgroups per CU of 4 waves -> 4 waves per SIMD in separate workgroups
  run(1, 4, it_long / 4, "hot");
  return 0;
}
