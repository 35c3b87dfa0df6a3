p='molgym_amd/csrc/backward.inc'
s=open(p).read()

# 1. template header + extra argument
old='''__global__ __launch_bounds__(64 * CGM_WAVES, 2) void k_catbuild_bwd_mfma(Lists L, const float* __restrict__ Acm,
                                                                      const float* __restrict__ Ecm, const float* __restrict__ Y,
                                                                      CatSrc dcat, EGrad dE, float* __restrict__ acm, CgTab tab,
                                                                      int TA, int TE) {'''
new='''// FUSED cat-mix adjoint (KS > 0, N = 4 KS real outputs of the level's atom cat-mix): the wave does not load its 6.2 KB slice
// of d_cat = dA_next W^H from memory (written by a column GEMM over all atoms: 31 MB per level on the SF6 mini-batch, a 20 us
// launch) -- it COMPUTES it.  The waves are channel-stationary (wave P serves channel P mod CH for its whole life), so the
// channel's part of W^H sits in registers as the A operands of 20 row tiles x KS reduction steps
// (A[tau'][o] = Mb_l[o][2 c W_l + tau'], loaded once per wave); per item only the atom's 25 rows of dA_next come from memory (B
// operands, through a sized resource: rows past 2l + 1 read 0), requested where the slice used to be requested; 20 KS MFMAs
// write the slice straight into its LDS buffer (D[tau' = 4q + r][m]: a lane holds four consecutive tau' of row m).
struct CgbMix {
  const float* dA[5];   // adjoint of the level's atom cat-mix output, per l [TA (2l+1)][N]
  const float* mb[5];   // the mix weights as rows per output: [N][ldb], columns channel-major (WPrep::perm_n)
  int ldb[5];
  int N;
};
template <int KS>
__global__ __launch_bounds__(64 * CGM_WAVES, 2) void k_catbuild_bwd_mfma_t(Lists L, const float* __restrict__ Acm,
                                                                        const float* __restrict__ Ecm, const float* __restrict__ Y,
                                                                        CatSrc dcat, EGrad dE, float* __restrict__ acm, CgTab tab,
                                                                        int TA, int TE, CgbMix mix) {'''
assert old in s
s=s.replace(old,new)

# 2. work order
old='''  int lwg = blockIdx.x >> 3;  // logical workgroup inside this XCD's share
  auto item_of = [&](int lw) { return ((int)(blockIdx.x & 7) * per_xcd + lw) * CGM_WAVES + wave; };'''
new='''  int lwg = blockIdx.x >> 3;  // logical workgroup inside this XCD's share
  // (fused form: channel-stationary waves -- wave P of the launch serves channel P mod CH and the atom slots P / CH, P / CH + K,
  // ... where K = the number of waves of that channel; ten consecutive waves work on the ten channels of one atom)
  const int fP = (int)blockIdx.x * CGM_WAVES + wave, fNW = (int)gridDim.x * CGM_WAVES;
  const int fC = fP % CH, fK = (fNW - fC + CH - 1) / CH;
  int fslot = fP / CH;
  auto item_of = [&](int lw) { return KS > 0 ? fslot * CH + fC : ((int)(blockIdx.x & 7) * per_xcd + lw) * CGM_WAVES + wave; };'''
assert old in s
s=s.replace(old,new)

# 3. slice_load -> also B-operand loads; weights
old='''  // first item of this wave
  int work = item_of(lwg);
  bool live = lwg < per_xcd && work < nwork;'''
new='''  // fused form: the channel's weights (once), the item's rows of dA_next (per item)
  constexpr int KSA = KS > 0 ? KS : 1;
  constexpr int FT0[6] = {0, 2, 6, 11, 16, 20};  // first row tile of part l: ceil(2 W_l / 16) = {2, 4, 5, 5, 4} tiles of 16 tau'
  float wA[KS > 0 ? 20 : 1][KSA];
  float bD[5][KSA];
  int voB = 0;
  if constexpr (KS > 0) {
#pragma unroll
    for (int l = 0; l < 5; ++l) {
      const int W2 = 2 * (2 * cgm_nblk(l) + 1);
      const float* mbl = mix.mb[l] + (size_t)q * mix.ldb[l] + W2 * fC + i;
#pragma unroll
      for (int g = FT0[l]; g < FT0[l + 1]; ++g) {
        const bool ok = 16 * (g - FT0[l]) + i < W2;
#pragma unroll
        for (int s2 = 0; s2 < KS; ++s2) wA[g][s2] = ok ? mbl[(size_t)(4 * s2) * mix.ldb[l] + 16 * (g - FT0[l])] : 0.f;
      }
    }
    voB = (i * mix.N + q) * 4;
  }
  auto mix_load = [&](int a) {  // B operands: dA_next[(a, l, m = i)][o = 4 s + q]; rows past 2l + 1 are past the resource
#pragma unroll
    for (int l = 0; l < 5; ++l) {
      const i32x4 rs = mg_rsrc(mix.dA[l] + (size_t)a * (2 * l + 1) * mix.N, (unsigned)((2 * l + 1) * mix.N * 4));
#pragma unroll
      for (int s2 = 0; s2 < KSA; ++s2) bD[l][s2] = mg_buffer_load_f32(rs, voB, 16 * s2, 0);
    }
  };
  // first item of this wave
  int work = item_of(lwg);
  bool live = KS > 0 ? fslot < TA : (lwg < per_xcd && work < nwork);'''
assert old in s
s=s.replace(old,new)
old='''    e0 = __builtin_amdgcn_readfirstlane(desc.z); a0 = __builtin_amdgcn_readfirstlane(desc.w);
    slice_load(a, c);
  }'''
new='''    e0 = __builtin_amdgcn_readfirstlane(desc.z); a0 = __builtin_amdgcn_readfirstlane(desc.w);
    if constexpr (KS > 0) mix_load(a); else slice_load(a, c);
  }'''
assert old in s
s=s.replace(old,new)

# 4. item top: slice from MFMA
old='''#pragma unroll
    for (int l = 0; l < 5; ++l) {
      const int W = 2 * cgm_nblk(l) + 1, SZ = (2 * l + 1) * W;
#pragma unroll
      for (int k = 0; k < SLK[l + 1] - SLK[l]; ++k)  // (the clamped lanes of the last load store the same value again)
        sl[cgm_slice_base(l) + min(lane + 64 * k, SZ - 1)] = {sv[SLK[l] + k].x, sv[SLK[l] + k].y};
    }'''
new='''    if constexpr (KS > 0) {
      float* const slf = reinterpret_cast<float*>(sl);
#pragma unroll
      for (int l = 0; l < 5; ++l) {
        const int W2 = 2 * (2 * cgm_nblk(l) + 1), nt = FT0[l + 1] - FT0[l];
        f32x4 acc[5];
#pragma unroll
        for (int g = 0; g < nt; ++g) acc[g] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s2 = 0; s2 < KS; ++s2)
#pragma unroll
          for (int g = 0; g < nt; ++g) acc[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(wA[FT0[l] + g][s2], bD[l][s2], acc[g], 0, 0, 0);
        if (i < 2 * l + 1) {  // register r of lane (m = i, q) of tile g is tau' = 16 g + 4 q + r of row m
          float* row = slf + 2 * cgm_slice_base(l) + i * W2 + 4 * q;
#pragma unroll
          for (int g = 0; g < nt; ++g) {
            if (16 * g + 16 <= W2) {
              *reinterpret_cast<float2*>(row + 16 * g) = {acc[g][0], acc[g][1]};
              *reinterpret_cast<float2*>(row + 16 * g + 2) = {acc[g][2], acc[g][3]};
            } else {  // the last tile of the part: tau' past the row's end belong to the next row
              if (16 * g + 4 * q < W2) *reinterpret_cast<float2*>(row + 16 * g) = {acc[g][0], acc[g][1]};
              if (16 * g + 4 * q + 2 < W2) *reinterpret_cast<float2*>(row + 16 * g + 2) = {acc[g][2], acc[g][3]};
            }
          }
        }
      }
    } else {
#pragma unroll
    for (int l = 0; l < 5; ++l) {
      const int W = 2 * cgm_nblk(l) + 1, SZ = (2 * l + 1) * W;
#pragma unroll
      for (int k = 0; k < SLK[l + 1] - SLK[l]; ++k)  // (the clamped lanes of the last load store the same value again)
        sl[cgm_slice_base(l) + min(lane + 64 * k, SZ - 1)] = {sv[SLK[l] + k].x, sv[SLK[l] + k].y};
    }
    }'''
assert old in s
s=s.replace(old,new)

# 5. next item
old='''    lwg += wg_stride;
    const int work_n = item_of(lwg);
    const bool more_items = lwg < per_xcd && work_n < nwork;'''
new='''    lwg += wg_stride;
    fslot += fK;
    const int work_n = item_of(lwg);
    const bool more_items = KS > 0 ? fslot < TA : (lwg < per_xcd && work_n < nwork);'''
assert old in s
s=s.replace(old,new)
old='''  if (more_items) slice_load(a_n, c_n);'''
new='''  if (more_items) { if constexpr (KS > 0) mix_load(a_n); else slice_load(a_n, c_n); }'''
assert old in s
s=s.replace(old,new)

# 6. launch site
old='''        ProfScope prof(s, "k_catbuild_bwd_mfma");
        hipLaunchKernelGGL(k_catbuild_bwd_mfma, dim3(cgm_grid_persist(TA * CH, 2)), dim3(64 * CGM_WAVES), 0, s, w.L, w.Acm[k], w.Ecm[k], w.Y,
                           dc, dE, w.d_Acm, g_cgtab[cur_device()], TA, TE);'''
new='''        ProfScope prof(s, "k_catbuild_bwd_mfma");
        CgbMix mx;
        memset(&mx, 0, sizeof(mx));
        if (mix_fused) {
          mx.N = 2 * P.atom_cout[k];
          for (int l = 0; l < 5; ++l) { mx.dA[l] = w.d_A[k + 1][l]; mx.mb[l] = w.atom[k][l].mb; mx.ldb[l] = w.atom[k][l].ldb; }
        }
        const dim3 cg_grid(cgm_grid_persist(TA * CH, 2)), cg_block(64 * CGM_WAVES);
#define CGB_GO(KS_) hipLaunchKernelGGL((k_catbuild_bwd_mfma_t<KS_>), cg_grid, cg_block, 0, s, w.L, w.Acm[k], w.Ecm[k], w.Y, dc, dE, w.d_Acm, \\
                                       g_cgtab[cur_device()], TA, TE, mx)
        switch (mix_fused ? mx.N / 4 : 0) {
          case 3: CGB_GO(3); break;
          case 4: CGB_GO(4); break;
          case 5: CGB_GO(5); break;
          case 6: CGB_GO(6); break;
          default: CGB_GO(0); break;
        }
#undef CGB_GO'''
assert old in s
s=s.replace(old,new)

# 7. skip the column GEMM when fused
old='''      RC(launch_dw(s, gw, 5));
      RC(launch_gemm(s, gx, 5));
      // CG aggregate / power adjoint'''
new='''      RC(launch_dw(s, gw, 5));
      // levels >= 1, N = 12 .. 24: d_cat is computed inside the CG adjoint kernel (k_catbuild_bwd_mfma_t<KS>); MG_CGB_FUSED=0: the GEMM
      const bool mix_fused = k >= 1 && cgb_mix_fused(2 * P.atom_cout[k]);
      if (!mix_fused) RC(launch_gemm(s, gx, 5));
      // CG aggregate / power adjoint'''
assert old in s
s=s.replace(old,new)
s=s.replace('''// ---- orchestration -----------------------------------------------------------------------------
// one Linear layer backward''','''static bool cgb_mix_fused(int N) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("MG_CGB_FUSED"); on = e ? atoi(e) : 1; }
  return on && N % 4 == 0 && N >= 12 && N <= 24 && (2 * CH) % 4 == 0;
}

// ---- orchestration -----------------------------------------------------------------------------
// one Linear layer backward''')
open(p,'w').write(s)
