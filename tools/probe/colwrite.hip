// Probe (GPU box): what the memory system takes for the OUTPUT stream of the wide column GEMMs (the cat-mix adjoints d_cat =
// dA_next W^T: 2.6 GB written per level at 2048 x canvas 40, k_gemm_mfma_cols_ws at 1.7 - 2.3 TB/s).
// build: hipcc --offload-arch=gfx950 -O3 tools/probe/colwrite.hip -o tools/probe/colwrite ; run: tools/probe/colwrite [mult]
//  A  linear 16-byte stores, grid-stride
//  B  the MFMA D-layout stored directly: lane (i = lane & 15, q = lane >> 4), register r -> row 4 q + r, column 16 t + i: one
//     instruction = 4 rows x 64 contiguous bytes (half a 128-byte line per row); a wave owns 16 rows and walks the column tiles
//  B2 the same, two ADJACENT column tiles per step (the two halves of each line back to back)
//  C  the tile re-laid through registers as if staged in LDS: lane writes 16 bytes, 16 lanes = 256 contiguous bytes of a row,
//     4 rows per instruction
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_linear(float4* p, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) p[i] = {1.f, 2.f, 3.f, (float)i};
}
template <int PAIR>
__global__ __launch_bounds__(256) void k_dlayout(float* X, int rows, int R, int ld) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, q = lane >> 4;
  const int ntiles = rows / 16;
  for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
    float* x0 = X + (size_t)(tile * 16 + 4 * q) * ld + i;
    for (int t = 0; t < R / 16; t += PAIR) {
#pragma unroll
      for (int u = 0; u < PAIR; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r) x0[(size_t)r * ld + 16 * (t + u)] = (float)(tile + r);
    }
  }
}
__global__ __launch_bounds__(256) void k_rowwise(float* X, int rows, int R, int ld) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c4 = lane & 15, rr = lane >> 4;
  const int ntiles = rows / 16;
  for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
    for (int t = 0; t < R / 64; ++t) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        *reinterpret_cast<float4*>(X + (size_t)(tile * 16 + rr + 4 * k) * ld + 64 * t + 4 * c4) = {1.f, 2.f, 3.f, (float)tile};
    }
  }
}
int main(int argc, char** argv) {
  const int mult = argc > 1 ? atoi(argv[1]) : 6;
  const int rows = 142560 * mult, R = 704, ld = 704;
  const size_t n = (size_t)rows * ld, n4 = n / 4;
  float* X; CK(hipMalloc(&X, n * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("rows %d x %d floats = %.0f MB\n", rows, ld, n * 4 / 1e6);
  auto timeit = [&](const char* name, auto launch) {
    float best = 1e9f; const int reps = 5;
    for (int r = 0; r < reps; ++r) {
      (void)hipEventRecord(e0, 0); launch(); (void)hipEventRecord(e1, 0); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (r) best = ms < best ? ms : best;
    }
    printf("%-44s best %.1f us  %.2f TB/s\n", name, best * 1e3, n * 4 / (best * 1e-3) / 1e12);
  };
  timeit("A linear float4 stores", [&] { hipLaunchKernelGGL(k_linear, dim3(2048), dim3(256), 0, 0, (float4*)X, n4); });
  timeit("B D-layout, 64 B per row and instruction", [&] { hipLaunchKernelGGL(k_dlayout<1>, dim3(2048), dim3(256), 0, 0, X, rows, R, ld); });
  timeit("B2 D-layout, adjacent tile pairs", [&] { hipLaunchKernelGGL(k_dlayout<2>, dim3(2048), dim3(256), 0, 0, X, rows, R, ld); });
  timeit("B4 D-layout, four adjacent tiles", [&] { hipLaunchKernelGGL(k_dlayout<4>, dim3(2048), dim3(256), 0, 0, X, rows, R, ld); });
  timeit("C row-wise 16-byte stores (256 B per row)", [&] { hipLaunchKernelGGL(k_rowwise, dim3(2048), dim3(256), 0, 0, X, rows, R, ld); });
  timeit("C row-wise, 1024 workgroups", [&] { hipLaunchKernelGGL(k_rowwise, dim3(1024), dim3(256), 0, 0, X, rows, R, ld); });
  return 0;
}
