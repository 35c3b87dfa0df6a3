// common.h -- shared device helpers for the gfx950 kernels (hand-written HIP, CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MAXL 4
#define NLM 25   // sum_l (2l+1), l <= 4
// channel counts of the Cormorant encoder: compile-time constants of a library build (tools/arg_parser.py:55-60 defaults);
// other values are other builds of the same sources (-DCH=.. -DCE=..: molgym_amd/_lib.py builds and loads them on demand)
#ifndef CH
#define CH 10    // num_channels_hidden
#endif
#ifndef CE
#define CE 4     // num_channels_per_element
#endif
// Several kernels cover the NLM * CH (atom, channel) items of an atom -- and the 2 * NLM * CE (row, output) pairs of the
// mixer -- with ONE 256-thread workgroup and no loop (k_dot's channel-major transpose, k_dot_bwd, k_catbuild0(_bwd), the
// fused heads): a build with more channels would silently leave the tail unwritten, so it does not compile.
static_assert(NLM * CH <= 256, "num_channels_hidden <= 10: one 256-thread workgroup covers the NLM * CH items of an atom");
static_assert(2 * NLM * CE <= 256, "num_channels_per_element <= 5: one 256-thread workgroup covers the 2 * NLM * CE mixer outputs");
#ifndef NLEV
#define NLEV 3   // num_cg_levels: a build parameter like CH / CE (arg_parser.py:56; 2 .. 4; molgym_amd/_lib.py builds the variants)
#endif
static_assert(NLEV >= 2 && NLEV <= 4, "num_cg_levels 2 .. 4");
#define NLEVA (NLEV < 3 ? 3 : NLEV)  // extent of the per-level arrays: the fused small-batch kernels (NLEV == 3 only) name levels 1, 2
#define NRADF 32 // radial features per level
#define NLEB 1730
#define LEB_STRIDE 51

// phase timestamps inside a kernel (debug builds only: -DMG_TS; tools/ts_heads.sh)
#ifndef MG_EXP
#define MG_EXP 0
#endif
#ifdef MG_TS
__device__ unsigned long long g_ts[128];
__device__ int g_ts_block;
__device__ unsigned long long g_wg_ts[3 * 256 * 4];  // heads: {forward start, forward end, backward start, backward end} of every workgroup
#define WGTS(role, b, k) do { if (threadIdx.x == 0 && (b) < 256) g_wg_ts[(((role) * 256) + (b)) * 4 + (k)] = wall_clock64(); } while (0)
#define TS(i) do { if (blockIdx.x == g_ts_block && blockIdx.y == 0 && threadIdx.x == 0) g_ts[i] = wall_clock64(); } while (0)
#define TSY(i) do { if (blockIdx.x == g_ts_block && threadIdx.x == 0) g_ts[i] = wall_clock64(); } while (0)
#else
#define TS(i) do { } while (0)
#define TSY(i) do { } while (0)
#define WGTS(role, b, k) do { } while (0)
#endif

// Workgroup barrier that waits for the wave's LDS traffic ONLY.  __syncthreads() is "s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier":
// in a kernel whose phases each end in fire-and-forget global stores / atomics it adds a full store round trip (1-2 us on a
// latency-bound launch) to every phase.  Use where the phases hand data over through LDS and nothing a wave wrote to global
// memory is read back by the workgroup (loads still in flight are waited for by the compiler at their first use).
__device__ __forceinline__ void wg_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct cf {  // complex float
  float r, i;
};
__host__ __device__ __forceinline__ cf cmul(cf a, cf b) { return {a.r * b.r - a.i * b.i, a.r * b.i + a.i * b.r}; }
__host__ __device__ __forceinline__ cf cmulc(cf a, cf b) {  // a * conj(b)
  return {a.r * b.r + a.i * b.i, a.i * b.r - a.r * b.i};
}
__host__ __device__ __forceinline__ void cmac(cf& acc, cf a, cf b) {
  acc.r = fmaf(a.r, b.r, acc.r);
  acc.r = fmaf(-a.i, b.i, acc.r);
  acc.i = fmaf(a.r, b.i, acc.i);
  acc.i = fmaf(a.i, b.r, acc.i);
}
__host__ __device__ __forceinline__ void cmacc(cf& acc, cf a, cf b) {  // acc += a * conj(b)
  acc.r = fmaf(a.r, b.r, acc.r);
  acc.r = fmaf(a.i, b.i, acc.r);
  acc.i = fmaf(a.i, b.r, acc.i);
  acc.i = fmaf(-a.r, b.i, acc.i);
}

__host__ __device__ __forceinline__ int lm_l(int idx) { return idx < 1 ? 0 : idx < 4 ? 1 : idx < 9 ? 2 : idx < 16 ? 3 : 4; }
__host__ __device__ __forceinline__ int lm_m(int idx, int l) { return idx - l * l - l; }

// Standard (Condon-Shortley, 'qm') complex spherical harmonics Y_l^m, l <= 4, of a UNIT
// vector; index l*l + m + l.  A zero vector gives Y_00 only (cormorant: x/|x| -> 0).
__host__ __device__ inline void ylm25(float x, float y, float z, bool nonzero, float* yr, float* yi) {
  yr[0] = 0.28209479177387814f;
  yi[0] = 0.f;
  if (!nonzero) {
    for (int k = 1; k < NLM; ++k) { yr[k] = 0.f; yi[k] = 0.f; }
    return;
  }
  const float c1r = x, c1i = y;
  const float c2r = x * x - y * y, c2i = 2.f * x * y;
  const float c3r = c2r * x - c2i * y, c3i = c2r * y + c2i * x;
  const float c4r = c2r * c2r - c2i * c2i, c4i = 2.f * c2r * c2i;
  const float z2 = z * z;
  float q;
  // l = 1
  yr[2] = 0.4886025119029199f * z; yi[2] = 0.f;
  q = -0.3454941494713355f;
  yr[3] = q * c1r; yi[3] = q * c1i; yr[1] = -q * c1r; yi[1] = q * c1i;
  // l = 2
  yr[6] = 0.6307831305050401f * 0.5f * (3.f * z2 - 1.f); yi[6] = 0.f;
  q = 0.2575161346821264f * (-3.f * z);
  yr[7] = q * c1r; yi[7] = q * c1i; yr[5] = -q * c1r; yi[5] = q * c1i;
  q = 0.1287580673410632f * 3.f;
  yr[8] = q * c2r; yi[8] = q * c2i; yr[4] = q * c2r; yi[4] = -q * c2i;
  // l = 3
  yr[12] = 0.7463526651802308f * 0.5f * (5.f * z2 - 3.f) * z; yi[12] = 0.f;
  q = 0.21545345607610045f * (-0.5f) * (15.f * z2 - 3.f);
  yr[13] = q * c1r; yi[13] = q * c1i; yr[11] = -q * c1r; yi[11] = q * c1i;
  q = 0.06813236509555216f * 15.f * z;
  yr[14] = q * c2r; yi[14] = q * c2i; yr[10] = q * c2r; yi[10] = -q * c2i;
  q = 0.02781492157551894f * (-15.f);
  yr[15] = q * c3r; yi[15] = q * c3i; yr[9] = -q * c3r; yi[9] = q * c3i;
  // l = 4
  yr[20] = 0.8462843753216345f * 0.125f * ((35.f * z2 - 30.f) * z2 + 3.f); yi[20] = 0.f;
  q = 0.18923493915151202f * (-2.5f) * (7.f * z2 - 3.f) * z;
  yr[21] = q * c1r; yi[21] = q * c1i; yr[19] = -q * c1r; yi[19] = q * c1i;
  q = 0.044603102903819275f * 7.5f * (7.f * z2 - 1.f);
  yr[22] = q * c2r; yi[22] = q * c2i; yr[18] = q * c2r; yi[18] = -q * c2i;
  q = 0.011920680675222404f * (-105.f) * z;
  yr[23] = q * c3r; yi[23] = q * c3i; yr[17] = -q * c3r; yi[17] = q * c3i;
  q = 0.004214597070904597f * 105.f;
  yr[24] = q * c4r; yi[24] = q * c4i; yr[16] = q * c4r; yi[16] = -q * c4i;
}

// sqrt(4 pi / (2l+1)): 'unit' normalisation used by the relative harmonics.
__host__ __device__ __forceinline__ float unit_norm(int l) {
  const float t[5] = {3.5449077018110318f, 2.0466534158929770f, 1.5853309190424043f, 1.3398491713813576f,
                      1.1816359006036772f};
  return t[l];
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Per-degree pointers / strides arrive as small arrays inside kernel arguments; indexing those with a LANE-VARYING l
// makes the compiler fetch them from the kernarg segment per lane (a dependent global load in front of every data
// load).  They are read once into scalars and chosen with compare/select chains instead.
struct Sel5P { const float *p0, *p1, *p2, *p3, *p4; };
struct Sel5M { float *p0, *p1, *p2, *p3, *p4; };
struct Sel5I { int v0, v1, v2, v3, v4; };
__device__ __forceinline__ const float* sel5(const Sel5P& s, int l) {
  return l == 0 ? s.p0 : l == 1 ? s.p1 : l == 2 ? s.p2 : l == 3 ? s.p3 : s.p4;
}
__device__ __forceinline__ float* sel5(const Sel5M& s, int l) {
  return l == 0 ? s.p0 : l == 1 ? s.p1 : l == 2 ? s.p2 : l == 3 ? s.p3 : s.p4;
}
__device__ __forceinline__ int sel5(const Sel5I& s, int l) {
  return l == 0 ? s.v0 : l == 1 ? s.v1 : l == 2 ? s.v2 : l == 3 ? s.v3 : s.v4;
}
#define SEL5P(arr) Sel5P{(arr)[0], (arr)[1], (arr)[2], (arr)[3], (arr)[4]}
#define SEL5M(arr) Sel5M{(arr)[0], (arr)[1], (arr)[2], (arr)[3], (arr)[4]}
#define SEL5I(arr) Sel5I{(arr)[0], (arr)[1], (arr)[2], (arr)[3], (arr)[4]}
