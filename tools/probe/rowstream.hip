// Probe (GPU box): what the memory system delivers for the X stream of the tall-skinny row GEMMs.
// build: hipcc --offload-arch=gfx950 -O3 tools/probe/rowstream.hip -o tools/probe/rowstream ; run: tools/probe/rowstream
//  A  linear 16-byte reads, grid-stride (the float4-copy pattern of the guide's 6.3 TB/s)
//  B  the MFMA A-operand pattern: lane (i = lane & 15, q = lane >> 4) reads X[row0 + i][k0 + 4 q .. + 3], a wave owns 32 rows and
//     walks k0 (16 rows x 64 bytes per instruction, rows LD floats apart) -- k_gemm_mfma_rows64 / rows_ws
//  C  whole rows per instruction: lane j reads X[row][4 j .. + 3] (1 KB contiguous per instruction), a wave owns 32 rows
//  each with the buffer just written by another kernel ("dirty") and after a 1 GB eviction sweep ("cold")
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_fill(float4* p, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) p[i] = {1.f, 2.f, 3.f, 4.f};
}
__global__ void k_linear(const float4* p, size_t n4, float* out) {
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) { float4 v = p[i]; s += v.x + v.y + v.z + v.w; }
  if (s == 12345.f) out[0] = s;
}
template <int UNROLL>
__global__ __launch_bounds__(256) void k_rows_mfma(const float* X, int rows, int R, int ld, float* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i = lane & 15, q = lane >> 4;
  const int ntiles = rows / 32;
  float s = 0.f;
  for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
    const float* x0 = X + (size_t)(tile * 32 + i) * ld + 4 * q;
    const float* x1 = x0 + (size_t)16 * ld;
    for (int k0 = 0; k0 < R; k0 += 16 * UNROLL) {
      float4 a[UNROLL], b[UNROLL];
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) { a[u] = *reinterpret_cast<const float4*>(x0 + k0 + 16 * u); b[u] = *reinterpret_cast<const float4*>(x1 + k0 + 16 * u); }
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) s += a[u].x + a[u].w + b[u].y + b[u].z;
    }
  }
  if (s == 12345.f) out[0] = s;
}
__global__ __launch_bounds__(256) void k_rows_contig(const float* X, int rows, int R, int ld, float* out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ntiles = rows / 32;
  float s = 0.f;
  for (int tile = blockIdx.x * 4 + wave; tile < ntiles; tile += gridDim.x * 4) {
    for (int r0 = 0; r0 < 32; r0 += 4) {
      float4 v[4][3];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const int k = 256 * c + 4 * lane;
          v[r][c] = k < R ? *reinterpret_cast<const float4*>(X + (size_t)(tile * 32 + r0 + r) * ld + k) : float4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) s += v[r][c].x + v[r][c].w;
    }
  }
  if (s == 12345.f) out[0] = s;
}
int main(int argc, char** argv) {
  // ~ the cfg3 l = 3 / 4 row blocks: 401 MB; `rowstream 6` = the cfg5 size (2.4 GB), where the launches last 0.5 - 3 ms
  const int mult = argc > 1 ? atoi(argv[1]) : 1;
  // (`rowstream 1 12768`: the 35 MB of one level's rows at the SF6 mini-batch -- 12750 rows of up to 704 floats)
  const int rows = argc > 2 ? atoi(argv[2]) : 142560 * mult, R = 704, ld = 704;
  const size_t n = (size_t)rows * ld, n4 = n / 4;
  float *X, *out; float4* evict;
  const size_t ev4 = (size_t)1 << 26;  // 1 GB
  CK(hipMalloc(&X, n * 4)); CK(hipMalloc(&out, 64)); CK(hipMalloc(&evict, ev4 * 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto timeit = [&](const char* name, int mode, auto launch) {
    float best = 1e9f, sum = 0.f; const int reps = 6;
    for (int r = 0; r < reps; ++r) {
      if (mode == 0) hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, (float4*)X, n4);             // dirty: just written
      else hipLaunchKernelGGL(k_fill, dim3(2048), dim3(256), 0, 0, evict, ev4);                           // cold: something else swept the caches
      hipEventRecord(e0, 0); launch(); hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (r) { best = ms < best ? ms : best; sum += ms; }
    }
    printf("%-34s %s  best %.1f us  mean %.1f us  %.2f TB/s (best)\n", name, mode == 0 ? "dirty" : "cold ", best * 1e3, sum / (reps - 1) * 1e3, n * 4 / (best * 1e-3) / 1e12);
  };
  printf("rows %d x %d floats = %.0f MB\n", rows, ld, n * 4 / 1e6);
  for (int mode = 0; mode < 2; ++mode) {
    timeit("A linear float4", mode, [&] { hipLaunchKernelGGL(k_linear, dim3(2048), dim3(256), 0, 0, (const float4*)X, n4, out); });
    timeit("B mfma pattern, 2 blocks / trip", mode, [&] { hipLaunchKernelGGL(k_rows_mfma<2>, dim3(1114), dim3(256), 0, 0, X, rows, R, ld, out); });
    timeit("B mfma pattern, 4 blocks / trip", mode, [&] { hipLaunchKernelGGL(k_rows_mfma<4>, dim3(1114), dim3(256), 0, 0, X, rows, R, ld, out); });
    timeit("B mfma pattern, 4 blk, 512 wg", mode, [&] { hipLaunchKernelGGL(k_rows_mfma<4>, dim3(512), dim3(256), 0, 0, X, rows, R, ld, out); });
    timeit("B mfma pattern, 11 blk, 2048 wg", mode, [&] { hipLaunchKernelGGL(k_rows_mfma<11>, dim3(2048), dim3(256), 0, 0, X, rows, R, ld, out); });
    timeit("C whole rows per instruction", mode, [&] { hipLaunchKernelGGL(k_rows_contig, dim3(1114), dim3(256), 0, 0, X, rows, R, ld, out); });
  }
  return 0;
}
