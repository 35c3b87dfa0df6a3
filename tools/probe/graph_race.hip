// build (GPU box): /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/probe/graph_race.hip -o tools/probe/graph_race
// probe (GPU box): is it safe to update the parameters of a hipGraphExec's kernel nodes while earlier launches of the SAME
// exec are still queued / running?  Each replay r writes (r + 0.5) into slot r from a slow kernel; the host updates and launches
// 200 replays back to back without synchronising.  Any slot that does not hold its own value means the arguments of a launch in
// flight were overwritten.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_slow(float* p, int idx, float v, int spin) {
  float acc = v;
  for (int i = 0; i < spin; ++i) acc = acc * 1.0000001f + 1e-9f;
  if (threadIdx.x == 0 && blockIdx.x == 0) p[idx] = v + (acc > 1e30f ? 1.f : 0.f);
}
int main() {
  const int N = 8, R = 200;
  float* d; CK(hipMalloc(&d, R * N * 4)); CK(hipMemset(d, 0, R * N * 4));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipGraph_t g; CK(hipGraphCreate(&g, 0));
  std::vector<hipGraphNode_t> nodes(N);
  int idx = 0, spin = 20000; float v = 0.f;
  for (int i = 0; i < N; ++i) {
    hipKernelNodeParams p = {};
    void* args[] = {&d, &idx, &v, &spin};
    p.func = (void*)k_slow; p.gridDim = dim3(2); p.blockDim = dim3(64); p.kernelParams = args;
    CK(hipGraphAddKernelNode(&nodes[i], g, i ? &nodes[i - 1] : nullptr, i ? 1 : 0, &p));
  }
  hipGraphExec_t ex; CK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
  for (int r = 0; r < R; ++r) {
    for (int i = 0; i < N; ++i) {
      hipKernelNodeParams p = {};
      idx = r * N + i; v = (float)idx + 0.5f;
      void* args[] = {&d, &idx, &v, &spin};
      p.func = (void*)k_slow; p.gridDim = dim3(2); p.blockDim = dim3(64); p.kernelParams = args;
      CK(hipGraphExecKernelNodeSetParams(ex, nodes[i], &p));
    }
    CK(hipGraphLaunch(ex, s));
  }
  CK(hipStreamSynchronize(s));
  std::vector<float> h(R * N);
  CK(hipMemcpy(h.data(), d, R * N * 4, hipMemcpyDeviceToHost));
  int bad = 0;
  for (int k = 0; k < R * N; ++k) if (h[k] != (float)k + 0.5f) { if (bad < 5) printf("slot %d holds %g\n", k, h[k]); ++bad; }
  printf("%d of %d slots wrong\n", bad, R * N);
  return 0;
}
