import re
# ---------- encoder.inc: generalise list / prep bodies ----------
p='molgym_amd/csrc/encoder.inc'
s=open(p).read()
s=s.replace("  if (b == 1023) {\n    if (oa + sa != TA || oe + se != TE) atomicExch(L.err, 2);","  if (b == (int)blockDim.x - 1) {  // (the last thread holds the totals; B <= blockDim.x)\n    if (oa + sa != TA || oe + se != TE) atomicExch(L.err, 2);")
s=s.replace('''__device__ __forceinline__ void prep_weights_body(const WPrepArgs& args) {
  if (args.zero_f) {
    const size_t nth = (size_t)gridDim.x * gridDim.y * blockDim.x;
    const size_t me = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;''','''// (bx, by) of a virtual gx x gy grid: the weight preparation may be a ROLE of a larger launch (k_level0_fwd<true>)
__device__ __forceinline__ void prep_weights_body(const WPrepArgs& args, int bx, int by, int gx, int gy) {
  if (args.zero_f) {
    const size_t nth = (size_t)gx * gy * blockDim.x;
    const size_t me = ((size_t)by * gx + bx) * blockDim.x + threadIdx.x;''')
s=s.replace('''  const WPrep& w = args.w[blockIdx.y];
  const int total = w.O * w.Q;''','''  const WPrep& w = args.w[by];
  const int total = w.O * w.Q;''')
s=s.replace('''  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
    const int o = t / w.Q, q = tau_perm(t % w.Q, w.perm_n);
    if (w.cplx) {
      const float2 v = {w.src[2 * t], w.src[2 * t + 1]};''','''  for (int t = bx * blockDim.x + threadIdx.x; t < total; t += gx * blockDim.x) {
    const int o = t / w.Q, q = tau_perm(t % w.Q, w.perm_n);
    if (w.cplx) {
      const float2 v = {w.src[2 * t], w.src[2 * t + 1]};''')
s=s.replace('''  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < total; t += gridDim.x * blockDim.x) {
    const int o = t % w.O, qs = t / w.O, q = tau_perm(qs, w.perm_n);''','''  for (int t = bx * blockDim.x + threadIdx.x; t < total; t += gx * blockDim.x) {
    const int o = t % w.O, qs = t / w.O, q = tau_perm(qs, w.perm_n);''')
s=s.replace("__global__ void k_prep_weights(WPrepArgs args) { prep_weights_body(args); }","__global__ void k_prep_weights(WPrepArgs args) { prep_weights_body(args, blockIdx.x, blockIdx.y, gridDim.x, gridDim.y); }")
s=s.replace("    return;\n  }\n  prep_weights_body(args);\n}","    return;\n  }\n  prep_weights_body(args, blockIdx.x, blockIdx.y, gridDim.x, gridDim.y);\n}")
open(p,'w').write(s)

# ---------- level0.inc ----------
p='molgym_amd/csrc/level0.inc'
s=open(p).read()
# args
s=s.replace('''  const float* rad_mb0; int rad_stride;   // radial Linear (k, l): rad_mb0 + (5k + l) * rad_stride, rows of NRADF
  const float* edge_mb[5]; int edge_ldb[5];
  const float* atom_mb[5]; int atom_ldb[5];''','''  const float* rad_mb0; int rad_stride;   // radial Linear (k, l): rad_mb0 + (5k + l) * rad_stride, rows of NRADF
  const float* edge_mb[5]; int edge_ldb[5];
  const float* atom_mb[5]; int atom_ldb[5];
  // FRONT form (k_level0_fwd<true>: the step's first launch, nothing prepared yet): the same weights straight from theta --
  // radial / input Linears are [out][in] there (= the rows above), a complex mix W (O, Q, 2) gives row 2o / 2o + 1 of its
  // real expansion as (wr, -wi) / (wi, wr) per input -- and the batch, from which every workgroup derives its own descriptor
  int in_w_off, edge_w_off[5], edge_K[5], atom_w_off[5], atom_K[5];
  int theta16;             // theta and every slot above 16-byte aligned: 16-byte weight loads
  int B, cfgTA, cfgTE, n_prep;''')
# runtime-aligned row loader + complex fix-up
s=s.replace('''struct L0Args {''','''// the same with the alignment known at run time only (rows of theta: parameter slots are 4-byte aligned in general)
template <int N>
__device__ __forceinline__ void l0_row_rt(const float* __restrict__ p, float (&w)[N], bool on, bool aligned16) {
  if (N % 4 == 0 && aligned16) l0_row<N>(p, w, on);
  else {
#pragma unroll
    for (int i = 0; i < N; ++i) w[i] = on ? p[i] : 0.f;
  }
}
// row of the real expansion of a complex mix from its (wr, wi) pairs: output parity 0 -> (wr, -wi), 1 -> (wi, wr)
template <int N>
__device__ __forceinline__ void l0_cplx_row(float (&w)[N], int parity) {
#pragma unroll
  for (int q = 0; q < N / 2; ++q) {
    const float wr = w[2 * q], wi = w[2 * q + 1];
    w[2 * q] = parity ? wi : wr;
    w[2 * q + 1] = parity ? wr : -wi;
  }
}

struct L0Args {''',1)
# kernel header -> template
old=s[s.index('__global__ __launch_bounds__(L0_T, 4) void k_level0_fwd(L0Args g, Lists L) {'):s.index('  // e: edge cat-mix (j mod 2, degree l, output o): few threads')]
new='''// FRONT = true: the step's FIRST launch.  Workgroups [0, TA): the atoms (natural order), each deriving its own descriptor from the
// charges (counts, prefix sums over the <= 256 samples: one batch of loads + a scan -- what the descriptor load cost) and reading
// its weights straight from theta; workgroup TA: the list build every later kernel reads (lists_small_body); the rest: the
// derived weights of the later kernels + the zero of the weight-gradient scratch (prep_weights_body).  Nothing in this launch
// waits for another launch: the separate list / weight-preparation launch (11.7 us at the head of every step) is gone.
template <bool FRONT>
__global__ __launch_bounds__(L0_T, 4) void k_level0_fwd(L0Args g, Lists L, WPrepArgs prep) {
  __shared__ L0Lds S;
  const int t = threadIdx.x;
  if constexpr (FRONT) {
    if ((int)blockIdx.x == g.cfgTA) {  // the lists
      if (t < 4) L.err[t] = 0;
      __syncthreads();
      lists_small_body(g.charges, g.B, g.N, g.cfgTA, g.cfgTE, L);
      return;
    }
    if ((int)blockIdx.x > g.cfgTA) {   // derived weights: 4 workgroups per matrix
      const int p = (int)blockIdx.x - g.cfgTA - 1;
      prep_weights_body(prep, p & 3, p >> 2, 4, g.n_prep);
      return;
    }
  }
  TSL_INIT;
  TSL0(64);
  int4 desc = {0, 0, 0, 0};
  int list_err = 0;
  int cq[16];
  if constexpr (FRONT) {
    // this thread's sample: all its charges in one batch of loads (N <= 16)
#pragma unroll
    for (int k = 0; k < 16; ++k) cq[k] = (t < g.B && k < g.N) ? g.charges[t * g.N + k] : 0;
  } else {
    desc = L.atom_desc[blockIdx.x];  // FIRST: loads return in order, and everything below waits for this one only
    list_err = L.err[0];  // inconsistent inputs (mg_cov_check raises them): the descriptors may be garbage -- touch nothing
  }
  const bool th16 = g.theta16 != 0;
  // ---- weight prefetch ----
  // c: radial Linear (level k, degree l) output o: its row of 32 weights
  const int c_kl = t / (2 * CH), c_o = t - c_kl * (2 * CH), c_k = c_kl / 5, c_l = c_kl - 5 * c_k;
  const bool c_on = t < 15 * 2 * CH;
  float wc[NRADF], c_bias = 0.f;
  if constexpr (FRONT) l0_row_rt<NRADF>(g.theta + g.rad0 + c_k * g.lvl_stride + 16 + c_l * (L0_RADW + 2 * CH) + c_o * NRADF, wc, c_on, th16);
  else l0_row<NRADF>(g.rad_mb0 + (size_t)c_kl * g.rad_stride + c_o * NRADF, wc, c_on);
  if (c_on) c_bias = g.theta[g.rad0 + c_k * g.lvl_stride + 16 + c_l * (L0_RADW + 2 * CH) + L0_RADW + c_o];
'''
s=s.replace(old,new)
# input-linear weights
s=s.replace('''  for (int z = 0; z < MG_MAX_Z; ++z) wbag[z] = (i_on && z < g.Z) ? g.in_mf[(size_t)(3 * g.Z + z) * g.in_ldf + i_o] : 0.f;
  if (i_on && g.in_bias) i_bias = g.in_bias[i_o];

  if (list_err != 0) return;
  const int a = desc.x, n = desc.y & 255, b = desc.y >> 8, e0 = desc.z, a0 = desc.w;
  const int F = 4 * g.Z;''','''  const int F = 4 * g.Z;
  // element [f][o] of the input Linear's weight: the transposed copy, or theta's [out][in] in the FRONT form
  auto in_w = [&](int f, int o) { return FRONT ? g.theta[g.in_w_off + o * F + f] : g.in_mf[(size_t)f * g.in_ldf + o]; };
#pragma unroll
  for (int z = 0; z < MG_MAX_Z; ++z) wbag[z] = (i_on && z < g.Z) ? in_w(3 * g.Z + z, i_o) : 0.f;
  if (i_on && g.in_bias) i_bias = g.in_bias[i_o];

  int a, n, b, e0, a0;
  if constexpr (FRONT) {
    // atoms per sample, inclusive scans of n and n^2 over the samples (thread = sample), then the sample of atom blockIdx.x
    __shared__ int s_n[MG_LISTS_SMALL_B], s_a0[MG_LISTS_SMALL_B], s_e0[MG_LISTS_SMALL_B], s_wa[L0_T / 64], s_we[L0_T / 64], s_bad;
    int nb = 0;
    bool gap = false, bad = false;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const bool real = cq[k] > 0;
      if (k < g.N) {
        if (real && gap) bad = true;
        if (real) ++nb; else gap = true;
      }
    }
    if (t == 0) s_bad = 0;
    const int lane = t & 63, wave = t >> 6;
    int sa = nb, se = nb * nb;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int ua = __shfl_up(sa, d), ue = __shfl_up(se, d);
      if (lane >= d) { sa += ua; se += ue; }
    }
    if (lane == 63) { s_wa[wave] = sa; s_we[wave] = se; }
    wg_lds_barrier();
    int oa = 0, oe = 0, ta = 0, te = 0;
#pragma unroll
    for (int k = 0; k < L0_T / 64; ++k) {
      if (k < wave) { oa += s_wa[k]; oe += s_we[k]; }
      ta += s_wa[k]; te += s_we[k];
    }
    if (t < g.B) { s_n[t] = nb; s_a0[t] = oa + sa - nb; s_e0[t] = oe + se - nb * nb; }
    if (bad) s_bad = 1;
    wg_lds_barrier();
    if (s_bad != 0 || ta != g.cfgTA || te != g.cfgTE) return;  // (the list workgroup raises the flags mg_cov_check reports)
    a = blockIdx.x;
    int lo = 0, hi = g.B - 1;  // last sample whose first atom is <= a (samples without atoms share their successor's start)
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (s_a0[mid] <= a) lo = mid; else hi = mid - 1;
    }
    b = lo; n = s_n[b]; a0 = s_a0[b]; e0 = s_e0[b] + (a - a0) * n;
  } else {
    if (list_err != 0) return;
    a = desc.x; n = desc.y & 255; b = desc.y >> 8; e0 = desc.z; a0 = desc.w;
  }''')
# one-hot rows in the input linear
s=s.replace('''      if (zi >= 0) {
        const float* m3 = g.in_mf + (size_t)(3 * zi) * g.in_ldf + i_o;
        acc = fmaf(1.f, m3[0], acc);
        acc = fmaf(x, m3[g.in_ldf], acc);
        acc = fmaf(x * x, m3[2 * g.in_ldf], acc);
      }''','''      if (zi >= 0) {
        acc = fmaf(1.f, in_w(3 * zi, i_o), acc);
        acc = fmaf(x, in_w(3 * zi + 1, i_o), acc);
        acc = fmaf(x * x, in_w(3 * zi + 2, i_o), acc);
      }''')
# edge weights
s=s.replace('''    const float* row = sel5(SEL5P(g.edge_mb), e_l) + (size_t)e_o * sel5(SEL5I(g.edge_ldb), e_l);
    float lo[2 * CH], hi[2 * CH];
    l0_row<2 * CH>(row, lo, e_on);
    l0_row<2 * CH>(row + 2 * CH, hi, e_on && e_l == 0);''','''    float lo[2 * CH], hi[2 * CH];
    if constexpr (FRONT) {
      const float* row = g.theta + sel5(SEL5I(g.edge_w_off), e_l) + (e_o >> 1) * sel5(SEL5I(g.edge_K), e_l);
      l0_row_rt<2 * CH>(row, lo, e_on, th16);
      l0_row_rt<2 * CH>(row + 2 * CH, hi, e_on && e_l == 0, th16);
      l0_cplx_row<2 * CH>(lo, e_o & 1);
      l0_cplx_row<2 * CH>(hi, e_o & 1);
    } else {
      const float* row = sel5(SEL5P(g.edge_mb), e_l) + (size_t)e_o * sel5(SEL5I(g.edge_ldb), e_l);
      l0_row<2 * CH>(row, lo, e_on);
      l0_row<2 * CH>(row + 2 * CH, hi, e_on && e_l == 0);
    }''')
# atom weights
s=s.replace('''  l0_row<2 * CH>(sel5(SEL5P(g.atom_mb), g_l) + (size_t)g_o * sel5(SEL5I(g.atom_ldb), g_l) + g_k0, wg, g_on);''','''  if constexpr (FRONT) {
    l0_row_rt<2 * CH>(g.theta + sel5(SEL5I(g.atom_w_off), g_l) + (g_o >> 1) * sel5(SEL5I(g.atom_K), g_l) + g_k0, wg, g_on, th16);
    l0_cplx_row<2 * CH>(wg, g_o & 1);
  } else {
    l0_row<2 * CH>(sel5(SEL5P(g.atom_mb), g_l) + (size_t)g_o * sel5(SEL5I(g.atom_ldb), g_l) + g_k0, wg, g_on);
  }''')
open(p,'w').write(s)
