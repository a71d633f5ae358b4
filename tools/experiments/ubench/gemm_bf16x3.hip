// Microbenchmark for VERDICT r4 item 8: a split-bf16 GEMM -- C[M,N] = A[M,K] . W[N,K]^T with fp32 activations in HBM, each fp32 value
// split into hi = bf16(x) and lo = bf16(x - hi), and the product taken as hi.hi + hi.lo + lo.hi on v_mfma_f32_32x32x16_bf16 with fp32
// accumulation (TERMS = 3), against plain bf16 (TERMS = 1: hi.hi only).  A is split on the fly while its k-tile goes from registers to
// LDS; W is split once on the host (weights: offline).  Structure: the engine's register-staged 128x128 GEMM reduced to its simplest form
// (one LDS stage, two barriers per k-tile, the next k-tile's global loads in flight under the MFMAs).  NOT part of the library.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/experiments/ubench/gemm_bf16x3.hip -o /tmp/gemm_bf16x3 && /tmp/gemm_bf16x3
// Prints, per shape: us per launch and fp32-equivalent TFLOP/s (2MNK / time) for TERMS = 3 and 1, and max|C - C_fp64| / max|C_fp64| over
// sampled rows for both (inputs uniform in [-1, 1)); the engine's own fp32 ring GEMM on the same shape is timed by gemm_shapes.py.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

#define CHECK(x)                                                                  \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));   \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

constexpr int BM = 128, BN = 128, BK = 32;   // k-tile: 32 elements = 128 B of fp32 per A row, 64 B of bf16 per plane row
constexpr int LDR = 80;                      // LDS row stride in bytes (64 + 16: conflict-free ds_read_b128 over 32 rows)
constexpr int PLANE = 128 * LDR;             // one 128-row plane
constexpr int LDC_S = BN + 4;

template <int TERMS>
__global__ __launch_bounds__(256) void gemm_split(const float* __restrict__ A, const __bf16* __restrict__ Whi,
                                                  const __bf16* __restrict__ Wlo, float* __restrict__ C, int M, int N, int K,
                                                  int n_ntiles, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* sAhi = smem;               // [128][LDR]
  char* sAlo = smem + PLANE;
  char* sWhi = smem + 2 * PLANE;
  char* sWlo = smem + 3 * PLANE;
  float* Cs = reinterpret_cast<float*>(smem);   // [128][132] fp32, epilogue only (67.6 KB: the launch asks for that much)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, r32 = lane & 31, kh = lane >> 5;
  const int nk = K / BK;
  // staging maps: A -- 8 lanes per 128-B row, 32 rows per pass, 4 passes; W planes -- 4 lanes per 64-B row, 64 rows per pass, 2 passes
  const int arow = tid >> 3, acol = (tid & 7) * 4;      // fp32 column
  const int wrow = tid >> 2, wcol = (tid & 3) * 8;      // bf16 column

  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int mt = tile / n_ntiles, m0 = mt * BM, n0 = (tile - mt * n_ntiles) * BN;
    f32x4 ra[4];
    bf16x8 rwh[2], rwl[2];
    auto gload = [&](int kt) {
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        int row = m0 + arow + 32 * p;
        row = row < M ? row : M - 1;
        ra[p] = *reinterpret_cast<const f32x4*>(A + (size_t)row * K + kt * BK + acol);
      }
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const size_t off = (size_t)(n0 + wrow + 64 * p) * K + kt * BK + wcol;
        rwh[p] = *reinterpret_cast<const bf16x8*>(Whi + off);
        if (TERMS > 1) rwl[p] = *reinterpret_cast<const bf16x8*>(Wlo + off);
      }
    };
    auto sstore = [&] {
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const bf16x4 hi = __builtin_convertvector(ra[p], bf16x4);
        *reinterpret_cast<bf16x4*>(sAhi + (arow + 32 * p) * LDR + acol * 2) = hi;
        if (TERMS > 1) {
          const f32x4 back = {(float)hi[0], (float)hi[1], (float)hi[2], (float)hi[3]};
          *reinterpret_cast<bf16x4*>(sAlo + (arow + 32 * p) * LDR + acol * 2) = __builtin_convertvector(ra[p] - back, bf16x4);
        }
      }
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        *reinterpret_cast<bf16x8*>(sWhi + (wrow + 64 * p) * LDR + wcol * 2) = rwh[p];
        if (TERMS > 1) *reinterpret_cast<bf16x8*>(sWlo + (wrow + 64 * p) * LDR + wcol * 2) = rwl[p];
      }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    gload(0);
    for (int kt = 0; kt < nk; ++kt) {
      sstore();
      __syncthreads();
      if (kt + 1 < nk) gload(kt + 1);   // in flight under the MFMAs
      const int abase = (wm * 64 + r32) * LDR + 16 * kh, bbase = (wn * 64 + r32) * LDR + 16 * kh;
#pragma unroll
      for (int s = 0; s < 2; ++s) {   // two 16-deep k-steps per k-tile; lane half kh holds k = 8kh .. 8kh+7 of each
        bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          ah[i] = *reinterpret_cast<const bf16x8*>(sAhi + abase + i * 32 * LDR + 32 * s);
          if (TERMS > 1) al[i] = *reinterpret_cast<const bf16x8*>(sAlo + abase + i * 32 * LDR + 32 * s);
          bh[i] = *reinterpret_cast<const bf16x8*>(sWhi + bbase + i * 32 * LDR + 32 * s);
          if (TERMS > 1) bl[i] = *reinterpret_cast<const bf16x8*>(sWlo + bbase + i * 32 * LDR + 32 * s);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            if (TERMS > 1) {   // the two small terms first, then the large one
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
            }
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
          }
      }
      __syncthreads();
    }
    // epilogue: accumulators -> LDS tile -> coalesced fp32 rows
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        float* cbase = Cs + (wm * 64 + i * 32 + 4 * kh) * LDC_S + wn * 64 + j * 32 + r32;
#pragma unroll
        for (int r = 0; r < 16; ++r) cbase[((r & 3) + 8 * (r >> 2)) * LDC_S] = acc[i][j][r];
      }
    __syncthreads();
    const int c0 = (tid & 31) * 4, rr = tid >> 5;
    for (int p = 0; p < 16; ++p) {
      const int row = rr + 8 * p, m = m0 + row;
      if (m < M) *reinterpret_cast<f32x4*>(C + (size_t)m * N + n0 + c0) = *reinterpret_cast<const f32x4*>(Cs + row * LDC_S + c0);
    }
    __syncthreads();
  }
}

static unsigned short bf16_rne(float f) {
  unsigned u;
  memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
static float bf16_f(unsigned short h) {
  unsigned u = (unsigned)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

template <int TERMS>
static float run(const float* dA, const __bf16* dWh, const __bf16* dWl, float* dC, int M, int N, int K, int reps) {
  const int n_ntiles = N / BN, ntiles = ((M + BM - 1) / BM) * n_ntiles;
  const size_t lds = (size_t)BM * LDC_S * 4;   // 67,584 B (>= the 4 planes' 40,960): two workgroups per CU
  static bool once = false;
  if (!once) {
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_split<TERMS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    once = true;
  }
  const int grid = ntiles < 512 ? ntiles : 512;
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(gemm_split<TERMS>, dim3(grid), dim3(256), lds, 0, dA, dWh, dWl, dC, M, N, K, n_ntiles, ntiles);
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  CHECK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(gemm_split<TERMS>, dim3(grid), dim3(256), lds, 0, dA, dWh, dWl, dC, M, N, K, n_ntiles, ntiles);
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  CHECK(hipGetLastError());
  return ms * 1000.f / reps;
}

int main() {
  const int shapes[][3] = {{6400, 1024, 512}, {3200, 1024, 512}, {3200, 512, 1024}, {3200, 2048, 1024}, {12800, 256, 512}, {51200, 128, 256}};
  for (const auto& sh : shapes) {
    const int M = sh[0], N = sh[1], K = sh[2];
    std::vector<float> A((size_t)M * K), W((size_t)N * K);
    srand(M + N + K);
    for (auto& v : A) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    for (auto& v : W) v = ((float)rand() / RAND_MAX * 2.f - 1.f) / sqrtf((float)K);
    std::vector<unsigned short> Wh(W.size()), Wl(W.size());
    for (size_t i = 0; i < W.size(); ++i) {
      Wh[i] = bf16_rne(W[i]);
      Wl[i] = bf16_rne(W[i] - bf16_f(Wh[i]));
    }
    float *dA, *dC;
    __bf16 *dWh, *dWl;
    CHECK(hipMalloc(&dA, A.size() * 4));
    CHECK(hipMalloc(&dC, (size_t)M * N * 4));
    CHECK(hipMalloc(&dWh, W.size() * 2));
    CHECK(hipMalloc(&dWl, W.size() * 2));
    CHECK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dWh, Wh.data(), W.size() * 2, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dWl, Wl.data(), W.size() * 2, hipMemcpyHostToDevice));
    // reference on sampled rows (first tile, a middle row block, the ragged end) in fp64
    std::vector<int> rows;
    for (int r = 0; r < 16; ++r) rows.push_back(r), rows.push_back(M / 2 + r), rows.push_back(M - 1 - r);
    std::vector<double> ref(rows.size() * (size_t)N);
    double refmax = 0;
    for (size_t ri = 0; ri < rows.size(); ++ri)
      for (int n = 0; n < N; ++n) {
        double s = 0;
        for (int k = 0; k < K; ++k) s += (double)A[(size_t)rows[ri] * K + k] * (double)W[(size_t)n * K + k];
        ref[ri * N + n] = s;
        refmax = fmax(refmax, fabs(s));
      }
    std::vector<float> Cc((size_t)M * N);
    double err[2];
    float us[2];
    for (int t = 0; t < 2; ++t) {
      us[t] = t == 0 ? run<3>(dA, dWh, dWl, dC, M, N, K, 200) : run<1>(dA, dWh, dWl, dC, M, N, K, 200);
      CHECK(hipMemcpy(Cc.data(), dC, Cc.size() * 4, hipMemcpyDeviceToHost));
      double e = 0;
      for (size_t ri = 0; ri < rows.size(); ++ri)
        for (int n = 0; n < N; ++n) e = fmax(e, fabs((double)Cc[(size_t)rows[ri] * N + n] - ref[ri * N + n]));
      err[t] = e / refmax;
    }
    const double fl = 2.0 * M * N * K;
    printf("M=%6d N=%5d K=%5d | bf16x3 %7.1f us %6.1f TF(fp32-equivalent) max|d|/max|ref| %.2e | plain bf16 %7.1f us %6.1f TF %.2e\n", M, N, K,
           us[0], fl / us[0] / 1e6, err[0], us[1], fl / us[1] / 1e6, err[1]);
    fflush(stdout);
    hipFree(dA), hipFree(dC), hipFree(dWh), hipFree(dWl);
  }
  return 0;
}
