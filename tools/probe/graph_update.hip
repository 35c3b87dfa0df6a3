// build (GPU box): /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/probe/graph_update.hip -o tools/probe/graph_update
// probe (GPU box): what does replaying a linear hipGraph of N kernel nodes cost on the host when EVERY node's parameters
// (grid + arguments) are updated before each launch (hipGraphExecKernelNodeSetParams), against N plain launches?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
struct Big { float* p; int n; int pad[60]; };
__global__ void k_a(Big a, float v) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < a.n) a.p[i] += v; }
__global__ void k_b(float* p, int n, float v) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 0.5f + v; }
int main() {
  const int N = 27, R = 200;
  float* d; CK(hipMalloc(&d, 1 << 20)); CK(hipMemset(d, 0, 1 << 20));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipGraph_t g; CK(hipGraphCreate(&g, 0));
  std::vector<hipGraphNode_t> nodes(N);
  Big a = {d, 1000, {0}}; float v = 1.f; int n = 1000;
  for (int i = 0; i < N; ++i) {
    hipKernelNodeParams p = {};
    void* argsA[] = {&a, &v}; void* argsB[] = {&d, &n, &v};
    p.func = (i & 1) ? (void*)k_b : (void*)k_a; p.gridDim = dim3(4); p.blockDim = dim3(256); p.kernelParams = (i & 1) ? argsB : argsA;
    CK(hipGraphAddKernelNode(&nodes[i], g, i ? &nodes[i - 1] : nullptr, i ? 1 : 0, &p));
  }
  hipGraphExec_t ex; CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ex, s)); CK(hipStreamSynchronize(s));
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
  // (1) plain launches
  auto t0 = now();
  for (int r = 0; r < R; ++r)
    for (int i = 0; i < N; ++i) {
      if (i & 1) hipLaunchKernelGGL(k_b, dim3(4), dim3(256), 0, s, d, n, v); else hipLaunchKernelGGL(k_a, dim3(4), dim3(256), 0, s, a, v);
    }
  auto t1 = now(); CK(hipStreamSynchronize(s)); auto t1b = now();
  printf("plain: %.1f us host per %d launches (%.2f us each); drain %.1f us\n", us(t0, t1) / R, N, us(t0, t1) / R / N, us(t1, t1b));
  // (2) replay without updates
  t0 = now();
  for (int r = 0; r < R; ++r) CK(hipGraphLaunch(ex, s));
  t1 = now(); CK(hipStreamSynchronize(s)); t1b = now();
  printf("graph replay, no update: %.1f us host per replay; drain %.1f us (GPU time per replay ~%.1f us)\n", us(t0, t1) / R, us(t1, t1b), us(t0, t1b) / R);
  // (3) replay with every node updated (grid changes too)
  t0 = now();
  double tup = 0;
  for (int r = 0; r < R; ++r) {
    auto u0 = now();
    for (int i = 0; i < N; ++i) {
      hipKernelNodeParams p = {};
      a.n = 900 + r; n = 900 + r; v = (float)r;
      void* argsA[] = {&a, &v}; void* argsB[] = {&d, &n, &v};
      p.func = (i & 1) ? (void*)k_b : (void*)k_a; p.gridDim = dim3(4 + (r & 3)); p.blockDim = dim3(256); p.kernelParams = (i & 1) ? argsB : argsA;
      CK(hipGraphExecKernelNodeSetParams(ex, nodes[i], &p));
    }
    tup += us(u0, now());
    CK(hipGraphLaunch(ex, s));
  }
  t1 = now(); CK(hipStreamSynchronize(s)); t1b = now();
  printf("graph replay, all %d nodes updated: %.1f us host per replay (updates %.1f us = %.2f us per node); total per replay incl. GPU %.1f us\n",
         N, us(t0, t1) / R, tup / R, tup / R / N, us(t0, t1b) / R);
  // (4) can the kernel FUNCTION of a node change?
  {
    hipKernelNodeParams p = {};
    void* argsB[] = {&d, &n, &v};
    p.func = (void*)k_b; p.gridDim = dim3(4); p.blockDim = dim3(256); p.kernelParams = argsB;
    hipError_t e = hipGraphExecKernelNodeSetParams(ex, nodes[0], &p);  // node 0 was k_a
    printf("changing a node's function: %s\n", hipGetErrorString(e));
    if (e == hipSuccess) { CK(hipGraphLaunch(ex, s)); CK(hipStreamSynchronize(s)); }
  }
  // (5) two execs of the same graph in flight on two streams
  hipGraphExec_t ex2; CK(hipGraphInstantiate(&ex2, g, nullptr, nullptr, 0));
  hipStream_t s2; CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  t0 = now();
  for (int r = 0; r < R; ++r) { CK(hipGraphLaunch(ex, s)); CK(hipGraphLaunch(ex2, s2)); }
  CK(hipStreamSynchronize(s)); CK(hipStreamSynchronize(s2));
  printf("two execs on two streams: %.1f us per pair of replays\n", us(t0, now()) / R);
  float h[4]; CK(hipMemcpy(h, d, 16, hipMemcpyDeviceToHost)); printf("check %g\n", h[0]);
  return 0;
}
