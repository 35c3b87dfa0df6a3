// molgym_hip.hip -- C-ABI entry points (include/molgym_hip.h) for the gfx950 PPO hot path.
// One translation unit: kernels live in the .inc files next to this one.
#include <mutex>
#include "state.inc"
#include "level0.inc"
#include "edge_level.inc"
#include "sampling.inc"
#include "heads_fused.inc"
#include "backward.inc"
#include "internal.inc"
#include "dists.inc"
#include "canvas.inc"

static bool g_tables_ready[MG_MAX_DEVICES];  // hipMemcpyToSymbol fills the CURRENT device's copy of a __constant__
static std::mutex g_tables_mutex;
static int ensure_tables() {
  const int dev = cur_device();
  if (g_tables_ready[dev]) return MG_OK;
  std::lock_guard<std::mutex> lock(g_tables_mutex);  // first use from several host threads at once
  if (g_tables_ready[dev]) return MG_OK;
  HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(d_cg_nblk), h_cg_nblk, sizeof(h_cg_nblk)));
  HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(d_cg_row_base), h_cg_row_base, sizeof(h_cg_row_base)));
  HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(d_cg_row_start), h_cg_row_start, sizeof(h_cg_row_start)));
  HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(d_cg_t_i1), h_cg_t_i1, sizeof(h_cg_t_i1)));
  HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(d_cg_t_i2), h_cg_t_i2, sizeof(h_cg_t_i2)));
  HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(d_cg_t_c), h_cg_t_c, sizeof(h_cg_t_c)));
  HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(d_cgT_start), h_cgT_start, sizeof(h_cgT_start)));
  HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(d_cgT_l), h_cgT_l, sizeof(h_cgT_l)));
  HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(d_cgT_blk), h_cgT_blk, sizeof(h_cgT_blk)));
  HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(d_cgT_c), h_cgT_c, sizeof(h_cgT_c)));
  HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(d_cgT_key), h_cgT_key, sizeof(h_cgT_key)));
  HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(d_cgI_start), h_cgI_start, sizeof(h_cgI_start)));
  HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(d_cgI_pk), h_cgI_pk, sizeof(h_cgI_pk)));
  HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(d_cgI_c), h_cgI_c, sizeof(h_cgI_c)));
  {  // the resolved lane-major tables of the one-wave-per-(atom, channel) CG kernels (copied into LDS once per workgroup)
    auto up = [](const void* src, size_t bytes, void** dst) {
      return hipMalloc(dst, bytes) == hipSuccess && hipMemcpy(*dst, src, bytes, hipMemcpyHostToDevice) == hipSuccess;
    };
    void *bt, *bq, *ft, *fq, *fl;
    static uint2 h_ftab[CG_FW_SLOTS * 64];
    for (int i = 0; i < CG_FW_SLOTS * 64; ++i) { h_ftab[i].x = h_cgFW_off[i]; memcpy(&h_ftab[i].y, &h_cgFW_c[i], 4); }
    // resolved adjoint tables: {offset, coefficient bits} word pairs, aggregate block then power block
    static uint2 h_btab[(CG_BK_SLOTS + CG_BP_SLOTS) * 64];
    static unsigned int h_bpos[(CG_KEY_NGRP + CG_PAIR_NGRP) * 64];
    for (int i = 0; i < CG_BK_SLOTS * 64; ++i) { h_btab[i].x = h_cgBK_off[i]; memcpy(&h_btab[i].y, &h_cgBK_c[i], 4); }
    for (int i = 0; i < CG_BP_SLOTS * 64; ++i) {
      h_btab[CG_BK_SLOTS * 64 + i].x = h_cgBP_off[i];
      memcpy(&h_btab[CG_BK_SLOTS * 64 + i].y, &h_cgBP_c[i], 4);
    }
    memcpy(h_bpos, h_cgBK_pos, sizeof(h_cgBK_pos));
    memcpy(h_bpos + CG_KEY_NGRP * 64, h_cgBP_pos, sizeof(h_cgBP_pos));
    if (!(up(h_btab, sizeof(h_btab), &bt) && up(h_bpos, sizeof(h_bpos), &bq) && up(h_ftab, sizeof(h_ftab), &ft) &&
          up(h_cgFW_pos, sizeof(h_cgFW_pos), &fq) && up(h_cgFW_lbm, sizeof(h_cgFW_lbm), &fl)))
      MG_FAIL(MG_EHIP, "uploading the CG term tables failed");
    g_cgtab[dev] = {(const uint2*)bt, (const unsigned int*)bq, (const uint2*)ft, (const unsigned int*)fq, (const unsigned int*)fl};
  }
  HIP_CHECK(hipDeviceSynchronize());
  g_tables_ready[dev] = true;
  return MG_OK;
}

extern "C" const char* mg_last_error(void) { return g_err; }

extern "C" int mg_gather_rows(int32_t nf, const void* const* src, void* const* dst, const int32_t* row_bytes, const int64_t* idx,
                              int32_t B, void* stream) {
  if (nf < 0 || nf > 8) MG_FAIL(MG_EINVAL, "mg_gather_rows: %d matrices (at most 8)", nf);
  if (B <= 0 || nf == 0) return MG_OK;
  GatherArgs a;
  memset(&a, 0, sizeof(a));
  a.nf = nf;
  for (int f = 0; f < nf; ++f) {
    if (row_bytes[f] <= 0 || row_bytes[f] % 4) MG_FAIL(MG_EINVAL, "mg_gather_rows: row of %d bytes", row_bytes[f]);
    a.src[f] = reinterpret_cast<const unsigned*>(src[f]);
    a.dst[f] = reinterpret_cast<unsigned*>(dst[f]);
    a.row_words[f] = row_bytes[f] / 4;
  }
  hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)B), dim3(64), 0, (hipStream_t)stream, a, reinterpret_cast<const long long*>(idx), B);
  LAUNCH_CHECK();
  return MG_OK;
}

extern "C" int mg_profile_enable(int on) {
  for (auto& sp : g_spans) { prof_flush(sp); sp.total_ms = 0; sp.count = 0; }
  g_prof_on = on ? 1 : 0;
  return MG_OK;
}
extern "C" int mg_profile_report(char* buf, size_t cap) {
  std::string out;
  char line[256];
  for (auto& sp : g_spans) {
    prof_flush(sp);
    snprintf(line, sizeof(line), "%s %.6f %ld\n", sp.name.c_str(), sp.total_ms, sp.count);
    out += line;
  }
  if (out.size() + 1 > cap) MG_FAIL(MG_ENOMEM, "profile report needs %zu bytes", out.size() + 1);
  memcpy(buf, out.c_str(), out.size() + 1);
  return MG_OK;
}
extern "C" int mg_abi_version(void) { return MG_ABI_VERSION; }
extern "C" int mg_cov_channels(int32_t* hidden, int32_t* per_element) {
  if (hidden) *hidden = CH;
  if (per_element) *per_element = CE;
  return MG_OK;
}

extern "C" int mg_cov_build_params(int32_t* hidden, int32_t* per_element, int32_t* maxl, int32_t* num_cg_levels) {
  if (hidden) *hidden = CH;
  if (per_element) *per_element = CE;
  if (maxl) *maxl = MAXL;
  if (num_cg_levels) *num_cg_levels = NLEV;
  return MG_OK;
}

extern "C" int mg_cov_num_params(const mg_cov_cfg* cfg, int64_t* num_params) {
  PLayout P;
  int rc = build_layout(cfg, &P);
  if (rc) return rc;
  *num_params = P.total;
  return MG_OK;
}

extern "C" int mg_cov_param_offsets(const mg_cov_cfg* cfg, int64_t* out, int32_t* n_slots) {
  PLayout P;
  int rc = build_layout(cfg, &P);
  if (rc) return rc;
  if (out) {
    for (size_t i = 0; i < P.slots.size(); ++i) out[i] = P.slots[i];
    out[P.slots.size()] = P.total;
  }
  *n_slots = (int32_t)P.slots.size();
  return MG_OK;
}

extern "C" int mg_cov_workspace_bytes(const mg_cov_cfg* cfg, size_t* bytes) {
  PLayout P;
  int rc = build_layout(cfg, &P);
  if (rc) return rc;
  WS w;
  rc = ws_build(cfg, P, nullptr, &w, nullptr);
  if (rc) return rc;
  *bytes = w.bytes;
  return MG_OK;
}

extern "C" int mg_cov_workspace_lookup(const mg_cov_cfg* cfg, const char* name, int64_t* off, int64_t* count) {
  PLayout P;
  int rc = build_layout(cfg, &P);
  if (rc) return rc;
  WS w;
  Arena ar;
  rc = ws_build(cfg, P, nullptr, &w, &ar);
  if (rc) return rc;
  for (size_t i = 0; i < ar.names.size(); ++i)
    if (ar.names[i] == name) {
      *off = (int64_t)(ar.offs[i] / sizeof(float));
      *count = (int64_t)ar.counts[i];
      return MG_OK;
    }
  MG_FAIL(MG_EINVAL, "no workspace entry named '%s'", name);
}

static int check_common(const mg_cov_cfg* c, const PLayout& P, const WS& w, size_t ws_bytes) {
  (void)P;
  if (c->B < 1) MG_FAIL(MG_EINVAL, "B must be >= 1");
  if (c->TA < 0 || c->TA > c->B * c->N) MG_FAIL(MG_EINVAL, "TA=%d inconsistent with B*N", c->TA);
  if (c->TE < c->TA || (long)c->TE > (long)c->TA * c->N) MG_FAIL(MG_EINVAL, "TE=%d inconsistent", c->TE);
  if (ws_bytes < w.bytes) MG_FAIL(MG_ENOMEM, "workspace %zu bytes < required %zu", ws_bytes, w.bytes);
  if (!(c->min_distance < c->max_distance)) MG_FAIL(MG_EINVAL, "min_distance must be < max_distance");
  return MG_OK;
}

struct ListsJob {  // the small list build riding on the weight-preparation launch (k_prep_lists)
  const int32_t* charges;
  int B, N, TA, TE;
};
static int prep_weights(hipStream_t s, const float* theta, WS& w, bool zero_scratch = false, const ListsJob* lj = nullptr) {
  std::vector<Lin*> all;
  for (int k = 0; k < NLEV; ++k)
    for (int l = 0; l < 5; ++l) { all.push_back(&w.rad[k][l]); all.push_back(&w.edge[k][l]); all.push_back(&w.atom[k][l]); }
  all.push_back(&w.lin_in);
  for (int l = 0; l < 5; ++l) all.push_back(&w.mix[l]);
  for (int m = 0; m < NMLP; ++m) { all.push_back(&w.mlp[m][0]); all.push_back(&w.mlp[m][1]); }
  for (size_t i0 = 0; i0 < all.size(); i0 += WPREP_MAX) {
    WPrepArgs a;
    memset(&a, 0, sizeof(a));
    const int n = (int)std::min((size_t)WPREP_MAX, all.size() - i0);
    for (int i = 0; i < n; ++i) {
      Lin* L = all[i0 + i];
      a.w[i] = {theta + L->w_off, L->mf, L->mb, L->O, L->Q, L->ldf, L->ldb, L->cplx, L->perm_n};
    }
    if (zero_scratch && i0 == 0) { a.zero_f = w.dwexp_all; a.zero_n = w.dwexp_floats; a.zero_i4 = w.L.err; }
    if (lj && i0 == 0) {
      a.zero_i4 = nullptr;  // cleared by the list row itself, before it may raise them
      hipLaunchKernelGGL(k_prep_lists, dim3(2, n + 1), dim3(1024), 0, s, a, n, lj->charges, lj->B, lj->N, lj->TA, lj->TE, w.L);
    } else {
      hipLaunchKernelGGL(k_prep_weights, dim3(8, n), dim3(256), 0, s, a);
    }
    LAUNCH_CHECK();
  }
  return MG_OK;
}

// [r5] the atom cat-mix of level k as the epilogue of k_catbuild_mfma (cg_mfma.inc: CgMix): MG_CATMIX_EPI=1.  Off by default: measured
// neutral on the SF6 mini-batch (0.3770 vs 0.3767 ms) and its float atomics cost the forward outputs their bit-for-bit repeatability
static bool catmix_epilogue(const WS& w, const PLayout& P, int k) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("MG_CATMIX_EPI"); on = e ? atoi(e) : 0; }
  if (!on || k < 1 || P.atom_cout[k] > 16) return false;  // (one output per lane quad)
  for (int l = 0; l < 5; ++l)
    if (w.atom[k][l].b_off >= 0 || !w.atom[k][l].cplx || w.atom[k][l].perm_n != 2 * kNblk[l] + 1) return false;
  return true;
}

struct SampleCtx {
  RngKey seed;
  int mode;
};
// actions is read-only unless smp != nullptr, in which case the sub-actions are drawn in place as the heads run.
static int cov_forward_impl(const mg_cov_cfg* c, const float* theta, const float* pos, const int32_t* charges,
                            const float* bags, float* actions, const float* leb, void* ws, size_t ws_bytes,
                            float* out, void* stream, const SampleCtx* smp, const PpoLossArgs* loss = nullptr,
                            bool* loss_fused = nullptr, int step_flags = 0) {
  PLayout P;
  int rc = build_layout(c, &P);
  if (rc) return rc;
  WS w;
  rc = ws_build(c, P, ws, &w, nullptr);
  if (rc) return rc;
  rc = check_common(c, P, w, ws_bytes);
  if (rc) return rc;
  rc = check_device_of(theta, "theta");
  if (rc) return rc;
  rc = check_device_of(ws, "workspace");
  if (rc) return rc;
  rc = ensure_tables();
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  const int B = c->B, N = c->N, Z = c->Z, TA = c->TA, TE = c->TE, W = c->W;
  side_policy(TE >= MG_SIDE_MIN_EDGES, (hipStream_t)stream);
#define RC(x) do { rc = (x); if (rc) return rc; } while (0)
  // the derived weight matrices (and the zero of the expanded weight-gradient scratch the backward accumulates
  // into) do not depend on the batch: side stream, beside the list / geometry kernels
  hipEvent_t weights_ready = nullptr;
  bool lists_done = false, front_lists = false;
  // MG_STEP_WEIGHTS_CURRENT: the derived weights of this workspace were written by an earlier step with the same theta (they sit
  // at batch-independent offsets, ws_build) and the expanded weight gradients are either zero (the last fold left them so) or
  // still accumulating (MG_STEP_DEFER_FOLD): neither is touched here
  const bool weights_current = (step_flags & MG_STEP_WEIGHTS_CURRENT) != 0;
  {
    hipStream_t ss = weights_current ? s : side_fork(s);
    if (ss == s) {
      // one launch: derived weights, zero of dwexp and of the error flags -- and, on small mini-batches, the list build
      const ListsJob lj = {charges, B, N, TA, TE};
      lists_done = (B <= MG_LISTS_SMALL_B && N <= 16);
      // ... unless the first fused kernel builds them itself (k_level0_fwd<true>: the list build as its workgroup 0, the atom
      // workgroups deriving their own descriptors): then NOTHING precedes that launch.  MG_FRONT_LISTS=0: the separate launch (A/B)
      static int front_on = -1;
      if (front_on < 0) { const char* e = getenv("MG_FRONT_LISTS"); front_on = e ? atoi(e) : 1; }
      front_lists = weights_current && lists_done && front_on && !smp && level0_fused(true, c, w);
      if (!weights_current) RC(prep_weights(s, theta, w, true, lists_done ? &lj : nullptr));
      else if (front_lists) {
      } else if (lists_done) {
        WPrepArgs none;
        memset(&none, 0, sizeof(none));
        hipLaunchKernelGGL(k_prep_lists, dim3(1, 1), dim3(1024), 0, s, none, 0, lj.charges, lj.B, lj.N, lj.TA, lj.TE, w.L);
        LAUNCH_CHECK();
      } else {  // (a kernel, not a memset: the step may be being recorded into a graph)
        WPrepArgs none;
        memset(&none, 0, sizeof(none));
        none.zero_i4 = w.L.err;
        hipLaunchKernelGGL(k_prep_weights, dim3(1, 1), dim3(64), 0, s, none);
        LAUNCH_CHECK();
      }
    } else {
      RC(prep_weights(ss, theta, w));
      HIP_CHECK(hipMemsetAsync(w.dwexp_all, 0, w.dwexp_floats * sizeof(float), ss));
      weights_ready = side_record();
      HIP_CHECK(hipMemsetAsync(w.L.err, 0, 4 * sizeof(int), s));
    }
  }
  if (lists_done) {
  } else if (B <= MG_LISTS_SMALL_B && N <= 16) {
    hipLaunchKernelGGL(k_lists_small, dim3(1), dim3(1024), 0, s, charges, B, N, TA, TE, w.L);
    LAUNCH_CHECK();
  } else {
    hipLaunchKernelGGL(k_prep_counts, dim3(1), dim3(256), 0, s, charges, B, N, TA, TE, w.L);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_fill_lists, dim3(B), dim3(64), 0, s, B, w.L);
    LAUNCH_CHECK();
  }
  int maxz = 0;
  ZsArr zs;
  for (int i = 0; i < 8; ++i) { zs.z[i] = i < Z ? c->zs[i] : -1; if (i < Z && c->zs[i] > maxz) maxz = c->zs[i]; }
  const float soft_rad = fmaxf(1e-3f, fminf(c->max_distance, 2.1f)), soft_width = 0.2f;
  InLinArgs ia = {TA, N, Z, zs, (float)maxz, c->bag_scale, charges, bags, w.lin_in.mf, w.lin_in.ldf,
                  w.lin_in.b_off >= 0 ? theta + w.lin_in.b_off : nullptr, 2 * CH, w.scal, w.A0};
  const unsigned n_in = (unsigned)((TA * 2 * CH + 255) / 256);
  bool input_done = false;
  // small mini-batches: inputs -> A[1] (geometry, input Linear, the 15 radial Linears and the whole of level 0) in ONE launch
  const bool fused0 = level0_fused(lists_done, c, w);
  if (fused0) {
    L0Args la;
    memset(&la, 0, sizeof(la));
    la.pos = pos; la.charges = charges; la.bags = bags; la.theta = theta;
    la.N = N; la.Z = Z; la.zs = zs; la.charge_scale = (float)maxz; la.bag_scale = c->bag_scale;
    la.rad0 = (int)P.rad_scales[0]; la.lvl_stride = (int)(P.rad_scales[1] - P.rad_scales[0]);
    la.soft_rad = soft_rad; la.soft_width = soft_width;
    la.in_mf = w.lin_in.mf; la.in_ldf = w.lin_in.ldf; la.in_bias = w.lin_in.b_off >= 0 ? theta + w.lin_in.b_off : nullptr;
    la.rad_mb0 = w.rad[0][0].mb; la.rad_stride = (int)(w.rad[0][1].mb - w.rad[0][0].mb);
    for (int kl = 0; kl < 15; ++kl)
      if (w.rad[kl / 5][kl % 5].mb != la.rad_mb0 + (size_t)kl * la.rad_stride || w.rad[kl / 5][kl % 5].ldb != NRADF)
        MG_FAIL(MG_EINVAL, "level-0 kernel: radial weights are not at a fixed stride");
    la.scal = w.scal; la.A0 = w.A0; la.r = w.r; la.em = w.em; la.Y = w.Y;
    for (int k = 0; k < 3; ++k) la.phi[k] = w.phi[k];
    la.ld_E0 = w.ld_e[1][0]; la.ld_rad1 = w.ld_e[1][0]; la.ld_rad2 = w.ld_e[2][0];
    for (int l = 0; l < 5; ++l) {
      la.edge_mb[l] = w.edge[0][l].mb; la.edge_ldb[l] = w.edge[0][l].ldb;
      la.atom_mb[l] = w.atom[0][l].mb; la.atom_ldb[l] = w.atom[0][l].ldb;
      la.cat_e0[l] = w.cat_e[0][l]; la.ld_e0[l] = w.ld_e[0][l]; la.rcol0[l] = w.rcol[0][l];
      la.E0[l] = w.cat_e[1][l];
      la.rad1[l] = w.cat_e[1][l] + w.rcol[1][l]; la.rad2[l] = w.cat_e[2][l] + w.rcol[2][l];
      la.cat_a0[l] = w.cat_a[0][l]; la.ld_a0[l] = w.ld_a[0][l];
      la.A1[l] = w.A[1][l];
      if (w.ld_e[1][l] != la.ld_E0 || w.ld_e[2][l] != la.ld_rad2)
        MG_FAIL(MG_EINVAL, "level-0 kernel: unexpected row strides");
    }
    la.B = B; la.cfgTA = TA; la.cfgTE = TE;
    ProfScope prof(s, "k_level0");
    if (front_lists && N <= 8) hipLaunchKernelGGL((k_level0_fwd<true, 8>), dim3(TA + 1), dim3(L0_T), 0, s, la, w.L);
    else if (front_lists) hipLaunchKernelGGL((k_level0_fwd<true, L0_MAXN>), dim3(TA + 1), dim3(L0_T), 0, s, la, w.L);
    else hipLaunchKernelGGL((k_level0_fwd<false, L0_MAXN>), dim3(TA), dim3(L0_T), 0, s, la, w.L);
    LAUNCH_CHECK();
    input_done = true;
  } else if (TE > 0) {
    GeomArgs ga = {TE, N, pos, theta, (int)P.rad_scales[0], (int)P.rad_phases[0], (int)(P.rad_scales[1] - P.rad_scales[0]),
                   soft_rad, soft_width, {w.r, w.em, w.Y, {}}};
    for (int k = 0; k < NLEV; ++k) ga.out.phi[k] = w.phi[k];
    const unsigned n_geom = (unsigned)((4 * TE + 255) / 256);
    if (!weights_ready && TA > 0 && !side_active()) {  // one stream: the input Linear rides on the geometry launch
      hipLaunchKernelGGL(k_geom_input, dim3(n_geom + n_in), dim3(256), 0, s, ga, ia, w.L, (int)n_geom);
      input_done = true;
    } else {
      hipLaunchKernelGGL(k_geom, dim3(n_geom), dim3(256), 0, s, ga, w.L);
    }
    LAUNCH_CHECK();
  }
  if (TE > 0 && TA > 0 && !fused0) {
    // the radial Linears of all three levels depend on the geometry only: side stream, beside the input Linear
    // and the first DotMatrix (they fill the radial columns of cat_e[k]; joined before the first edge cat-mix)
    hipStream_t ss = side_fork(s);
    if (!w.shared_dot) {
      GemmG gr[5 * NLEV];  // one launch for the NLEV x 5 radial Linears
      for (int k = 0; k < NLEV; ++k)
        for (int l = 0; l < 5; ++l)
          gr[5 * k + l] = fwd_group(w.rad[k][l], theta, w.phi[k], NRADF, w.cat_e[k][l] + w.rcol[k][l], w.ld_e[k][l], TE,
                                    0, nullptr);
      RC(launch_gemm(ss, gr, 5 * NLEV));
    } else {
    SxArgs ra;  // one launch for the 3 x 5 radial Linears: the five of a level share its radial basis
    memset(&ra, 0, sizeof(ra));
    ra.rows = TE;
    for (int k = 0; k < NLEV; ++k) {
      SxSet& S = ra.set[k];
      S.Xs = w.phi[k]; S.ldxs = NRADF; S.Rs = NRADF; S.ngroups = 5;
      for (int l = 0; l < 5; ++l) {
        const Lin& R = w.rad[k][l];
        SxG& g = S.g[l];
        g.M = R.mf; g.row_s = 0; g.ldm = R.ldf; g.N = R.N;
        g.bias = R.b_off >= 0 ? theta + R.b_off : nullptr;
        g.Y = w.cat_e[k][l] + w.rcol[k][l]; g.ldy = w.ld_e[k][l];
      }
    }
    RC(launch_sx(ss, ra, NLEV));
    }
  }
  if (TA > 0 && !input_done) {
    stream_wait(s, weights_ready);
    weights_ready = nullptr;
    hipLaunchKernelGGL(k_input_linear, dim3(n_in), dim3(256), 0, s, ia, w.L);
    LAUNCH_CHECK();
  }
  for (int k = fused0 ? 1 : 0; k < NLEV && TA > 0; ++k) {
    // --- edge level k (cormorant EdgeLevel: DotMatrix, cat-mix, soft mask) ---
    const bool fusedE = k >= 1 && edge_level_fused(fused0, N);
    if (fusedE) {
      ELArgs ea;
      memset(&ea, 0, sizeof(ea));
      for (int l = 0; l < 5; ++l) {
        ea.A[l] = w.A[k][l]; ea.row[l] = w.cat_e[k][l];
        ea.out[l] = (k < NLEV - 1) ? w.cat_e[k + 1][l] : w.Elast[l];
        ea.mb[l] = w.edge[k][l].mb;
        if (w.ld_e[k][l] != EL_K || w.edge[k][l].ldb != w.edge[k][0].ldb || w.edge[k][l].K != EL_K)
          MG_FAIL(MG_EINVAL, "edge-level kernel: unexpected row layout");
      }
      ea.ld_row = EL_K; ea.ld_out = (k < NLEV - 1) ? w.ld_e[k + 1][0] : 2 * CH; ea.ldb = w.edge[k][0].ldb;
      ea.em = w.em; ea.Acm = w.Acm[k]; ea.TA = TA; ea.N = N;
      if (!smp && catmix_epilogue(w, P, k)) {
        for (int l = 0; l < 5; ++l) ea.zA[l] = w.A[k + 1][l];
        ea.zld = 2 * P.atom_cout[k];
      }
      ProfScope prof(s, "k_edge_level");
      hipLaunchKernelGGL(k_edge_fwd, dim3(TA), dim3(EL_T), el_fwd_lds_bytes(N), s, ea, w.L);
      LAUNCH_CHECK();
    } else if (k == 0) {
      hipLaunchKernelGGL(k_dot0, dim3((TE * CH + 255) / 256), dim3(256), 0, s, TE, w.L, w.A0, w.cat_e[0][0],
                         w.ld_e[0][0], w.dcol[0]);
    } else {
      APtrs A;
      for (int l = 0; l < 5; ++l) A.p[l] = w.A[k][l];
      A.C = CH;
      DotDst d;
      if (w.shared_dot) {
        for (int l = 0; l < 5; ++l) { d.p[l] = w.dotbuf[k]; d.ld[l] = 10 * CH; }
        d.col = 0;
        d.nparts = 1;
      } else {
        for (int l = 0; l < 5; ++l) { d.p[l] = w.cat_e[k][l]; d.ld[l] = w.ld_e[k][l]; }
        d.col = w.dcol[k];
        d.nparts = 5;
      }
      const int n_dot = (TE * 5 * CH + 255) / 256;
      hipLaunchKernelGGL(k_dot, dim3(n_dot + TA), dim3(256), 0, s, TE, w.L, A, d, w.Acm[k], TA, n_dot);
    }
    LAUNCH_CHECK();
    if (k == 0) side_join(s);  // radial columns of every level are in place
    if (fusedE) {
    } else if (k == 0 || !w.shared_dot) {
      GemmG ge[5];
      for (int l = 0; l < 5; ++l) {
        float* Ek = (k < NLEV - 1) ? w.cat_e[k + 1][l] : w.Elast[l];
        const int ldE = (k < NLEV - 1) ? w.ld_e[k + 1][l] : 2 * CH;
        ge[l] = fwd_group(w.edge[k][l], theta, w.cat_e[k][l], w.ld_e[k][l], Ek, ldE, TE, 0, w.em);
      }
      RC(launch_gemm(s, ge, 5));  // (level 0: l = 0 has a different reduction width; the MFMA row form takes both)
    } else {
      // weight rows of the edge mix: [previous edge net (2 CH) | dot block (10 CH) | radial (2 CH)]
      SxArgs ea;
      memset(&ea, 0, sizeof(ea));
      ea.rows = TE;
      ea.rowscale = w.em;
      SxSet& S = ea.set[0];
      S.Xs = w.dotbuf[k]; S.ldxs = 10 * CH; S.Rs = 10 * CH; S.ngroups = 5;
      for (int l = 0; l < 5; ++l) {
        const Lin& E = w.edge[k][l];
        SxG& g = S.g[l];
        g.M = E.mf; g.ldm = E.ldf; g.N = E.N;
        g.row_p0 = 0; g.Rp0 = 2 * CH;
        g.row_s = 2 * CH;
        g.row_p1 = 12 * CH; g.Rp1 = 2 * CH;
        g.Xp = w.cat_e[k][l]; g.ldxp = w.ld_e[k][l];
        g.bias = E.b_off >= 0 ? theta + E.b_off : nullptr;
        g.Y = (k < NLEV - 1) ? w.cat_e[k + 1][l] : w.Elast[l];
        g.ldy = (k < NLEV - 1) ? w.ld_e[k + 1][l] : 2 * CH;
      }
      RC(launch_sx(s, ea, 1));
    }
    // --- atom level k (CG aggregate, CG power, cat-mix) ---
    EPtrs E;
    for (int l = 0; l < 5; ++l) {
      E.p[l] = (k < NLEV - 1) ? w.cat_e[k + 1][l] : w.Elast[l];
      E.ld[l] = (k < NLEV - 1) ? w.ld_e[k + 1][l] : 2 * CH;
    }
    CatDst cd;
    for (int l = 0; l < 5; ++l) { cd.p[l] = w.cat_a[k][l]; cd.ld[l] = w.ld_a[k][l]; }
    if (k == 0) {
      ProfScope prof(s, "k_catbuild0");
      hipLaunchKernelGGL(k_catbuild0, dim3(TA), dim3(256), 0, s, w.L, w.A0, E, w.Y, cd);
    } else {
      APtrs A;
      for (int l = 0; l < 5; ++l) A.p[l] = w.A[k][l];
      A.C = CH;
      ProfScope prof(s, "k_catbuild_mfma");
      CgMix mx;
      memset(&mx, 0, sizeof(mx));
      if (fusedE && !smp && catmix_epilogue(w, P, k)) {  // the atom cat-mix as the kernel's epilogue (A[k + 1] zeroed by k_edge_fwd above)
        for (int l = 0; l < 5; ++l) { mx.mf[l] = w.atom[k][l].mf; mx.ldf[l] = w.atom[k][l].ldf; mx.out[l] = w.A[k + 1][l]; }
        mx.ldo = 2 * P.atom_cout[k]; mx.nout = P.atom_cout[k];
        hipLaunchKernelGGL((k_catbuild_mfma<true, false>), dim3(cgm_grid_persist(TA * CH, 2)), dim3(64 * CGM_WAVES), 0, s, w.L, w.Acm[k], E,
                           w.Ecm[k], w.Y, cd, g_cgtab[cur_device()], TA, TE, mx);
        LAUNCH_CHECK();
        continue;
      }
      if (cgm_chunked(TA * CH))
        hipLaunchKernelGGL((k_catbuild_mfma<false, true>), dim3(cgm_grid_persist(TA * CH, 2)), dim3(64 * CGM_WAVES), 0, s, w.L, w.Acm[k], E, w.Ecm[k],
                           w.Y, cd, g_cgtab[cur_device()], TA, TE, mx);
      else
        hipLaunchKernelGGL((k_catbuild_mfma<false, false>), dim3(cgm_grid_persist(TA * CH, 2)), dim3(64 * CGM_WAVES), 0, s, w.L, w.Acm[k], E, w.Ecm[k],
                           w.Y, cd, g_cgtab[cur_device()], TA, TE, mx);
    }
    LAUNCH_CHECK();
    GemmG ga[5];
    for (int l = 0; l < 5; ++l)
      ga[l] = fwd_group(w.atom[k][l], theta, w.cat_a[k][l], w.ld_a[k][l], w.A[k + 1][l], 2 * P.atom_cout[k],
                        TA * (2 * l + 1), 0, nullptr);
    RC(launch_gemm(s, ga, 5));
  }
  // --- heads ---
  stream_wait(s, weights_ready);  // only still pending for a batch of empty canvases
  const int Co = P.Co, nlat = P.nlat;
  APtrs A3;
  for (int l = 0; l < 5; ++l) A3.p[l] = w.A[NLEV][l];
  A3.C = Co;
  if (!smp && !use_staged_heads(c->W)) {  // action evaluation: all heads in one launch (heads_fused.inc)
    HeadDims HD;
    HeadW HW;
    HeadBuf HB;
    make_head_args(c, P, w, theta, &HD, &HW, &HB);
    ProfScope prof(s, "k_heads_fwd");
    hipLaunchKernelGGL(k_heads_fwd, dim3(B, 3), dim3(256), head_smem_bytes(nlat, P.nlatE), s, HD, w.L, HW, HB, g_cgtab[cur_device()], A3, actions,
                       bags, leb, out, loss ? *loss : PpoLossArgs{});
    LAUNCH_CHECK();
    if (loss && loss_fused) *loss_fused = true;
    return MG_OK;
  }
  if (TA > 0) {
    hipLaunchKernelGGL(k_scalars, dim3((TA * Co + 255) / 256), dim3(256), 0, s, TA, Co, A3, w.inv);
    LAUNCH_CHECK();
    GemmG g2[2] = {fwd_group(w.mlp[MLP_FOCUS][0], theta, w.inv, nlat, w.hF, W, TA, 1, nullptr),
                   fwd_group(w.mlp[MLP_TRANS][0], theta, w.inv, nlat, w.hT, W, TA, 1, nullptr)};
    RC(launch_gemm(s, g2, 2));
    GemmG gf = fwd_group(w.mlp[MLP_FOCUS][1], theta, w.hF, W, w.logitF, 1, TA, 0, nullptr);
    RC(launch_gemm(s, &gf, 1));
    GemmG gt = fwd_group(w.mlp[MLP_TRANS][1], theta, w.hT, W, w.trans, W, TA, 0, nullptr);
    RC(launch_gemm(s, &gt, 1));
  }
  float* parts = w.parts;
  if (smp) {
    hipLaunchKernelGGL(k_sample_focus, dim3((B + 63) / 64), dim3(64), 0, s, B, w.L, w.logitF, smp->seed, smp->mode,
                       actions);
    LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(k_focus_head, dim3((B + 63) / 64), dim3(64), 0, s, B, w.L, w.logitF, actions, parts, parts + 4 * B,
                     w.fidx);
  LAUNCH_CHECK();
  EcovDst ec;
  for (int l = 0; l < 5; ++l) ec.p[l] = w.ecov[l];
  hipLaunchKernelGGL(k_gather_focus, dim3(B), dim3(256), 0, s, B, Co, nlat, w.fidx, actions, w.inv, A3, w.finv, ec);
  LAUNCH_CHECK();
  {
    GemmG g = fwd_group(w.mlp[MLP_ELEMENT][0], theta, w.finv, nlat, w.hE, W, B, 1, nullptr);
    RC(launch_gemm(s, &g, 1));
    g = fwd_group(w.mlp[MLP_ELEMENT][1], theta, w.hE, W, w.logitE, Z, B, 0, nullptr);
    RC(launch_gemm(s, &g, 1));
  }
  if (smp) {  // draw the element, then redo the channel selection that depends on it
    hipLaunchKernelGGL(k_sample_element, dim3((B + 63) / 64), dim3(64), 0, s, B, Z, w.logitE, bags, smp->seed,
                       smp->mode, actions);
    LAUNCH_CHECK();
    hipLaunchKernelGGL(k_gather_focus, dim3(B), dim3(256), 0, s, B, Co, nlat, w.fidx, actions, w.inv, A3, w.finv, ec);
    LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(k_element_head, dim3((B + 63) / 64), dim3(64), 0, s, B, Z, w.logitE, bags, actions, parts + B,
                     parts + 5 * B);
  LAUNCH_CHECK();
  {
    APtrs Ae;
    for (int l = 0; l < 5; ++l) Ae.p[l] = w.ecov[l];
    Ae.C = CE;
    hipLaunchKernelGGL(k_scalars, dim3((B * CE + 255) / 256), dim3(256), 0, s, B, CE, Ae, w.einv);
    LAUNCH_CHECK();
    GemmG g = fwd_group(w.mlp[MLP_D][0], theta, w.einv, P.nlatE, w.hD, W, B, 1, nullptr);
    RC(launch_gemm(s, &g, 1));
    g = fwd_group(w.mlp[MLP_D][1], theta, w.hD, W, w.dout, 2 * c->G, B, 0, nullptr);
    RC(launch_gemm(s, &g, 1));
    const float half_w = (c->max_distance - c->min_distance) / 2, center = (c->max_distance + c->min_distance) / 2;
    if (smp) {
      hipLaunchKernelGGL(k_sample_gmm, dim3((B + 63) / 64), dim3(64), 0, s, B, c->G, w.dout, theta + P.logstd, half_w,
                         center, smp->seed, smp->mode, actions);
      LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_gmm, dim3((B + 63) / 64), dim3(64), 0, s, B, c->G, w.dout, theta + P.logstd, actions, half_w,
                       center, parts + 2 * B);
    LAUNCH_CHECK();
  }
  {
    CatDst cm;
    for (int l = 0; l < 5; ++l) { cm.p[l] = w.cat_m[l]; cm.ld[l] = w.ld_m[l]; }
    hipLaunchKernelGGL(k_mixer_cat, dim3((B * CE * (CG_NROWS + NLM) + 255) / 256), dim3(256), 0, s, B, actions, ec, cm);
    LAUNCH_CHECK();
    GemmG gm[5];
    for (int l = 0; l < 5; ++l)
      gm[l] = fwd_group(w.mix[l], theta, w.cat_m[l], w.ld_m[l], w.cond[l], 2 * CE, B * (2 * l + 1), 0, nullptr);
    RC(launch_gemm(s, gm, 5));
    CondPtrs cp;
    for (int l = 0; l < 5; ++l) cp.p[l] = w.cond[l];
    if (smp) {
      hipLaunchKernelGGL(k_sample_so3, dim3(B), dim3(256), 0, s, B, w.L, cp, c->has_beta, c->beta, smp->seed, smp->mode,
                         actions);
      LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_so3, dim3(B), dim3(256), 0, s, B, w.L, cp, actions, leb, c->has_beta, c->beta, parts + 3 * B,
                       w.logz);
    LAUNCH_CHECK();
  }
  {
    hipLaunchKernelGGL(k_value_sum, dim3((B * W + 255) / 256), dim3(256), 0, s, B, W, w.L, w.trans, w.vfeat);
    LAUNCH_CHECK();
    GemmG g = fwd_group(w.mlp[MLP_V][0], theta, w.vfeat, W, w.hV, W, B, 1, nullptr);
    RC(launch_gemm(s, &g, 1));
    g = fwd_group(w.mlp[MLP_V][1], theta, w.hV, W, out + 2 * B, 1, B, 0, nullptr);
    RC(launch_gemm(s, &g, 1));
  }
  hipLaunchKernelGGL(k_finalize, dim3((B + 63) / 64), dim3(64), 0, s, B, parts, out);
  LAUNCH_CHECK();
#undef RC
  return MG_OK;
}

extern "C" int mg_cov_forward(const mg_cov_cfg* c, const float* theta, const float* pos, const int32_t* charges,
                              const float* bags, const float* actions, const float* leb, void* ws, size_t ws_bytes,
                              float* out, void* stream) {
  return cov_forward_impl(c, theta, pos, charges, bags, const_cast<float*>(actions), leb, ws, ws_bytes, out, stream,
                          nullptr);
}

extern "C" int mg_cov_sample_ids(const mg_cov_cfg* c, const float* theta, const float* pos, const int32_t* charges,
                                 const float* bags, const float* leb, uint64_t seed, int32_t sample_base, int32_t sample_stride,
                                 int32_t mode, void* ws, size_t ws_bytes, float* actions_out, float* out, void* stream) {
  if (mode != SAMPLE_TRAIN && mode != SAMPLE_EVAL) MG_FAIL(MG_EINVAL, "mode must be 1 (sample) or 2 (argmax)");
  if (sample_base < 0 || sample_stride < 1) MG_FAIL(MG_EINVAL, "sample ids base %d stride %d", sample_base, sample_stride);
  HIP_CHECK(hipMemsetAsync(actions_out, 0, (size_t)c->B * 6 * sizeof(float), (hipStream_t)stream));
  SampleCtx smp = {{seed, sample_base, sample_stride}, mode};
  return cov_forward_impl(c, theta, pos, charges, bags, actions_out, leb, ws, ws_bytes, out, stream, &smp);
}

extern "C" int mg_cov_sample(const mg_cov_cfg* c, const float* theta, const float* pos, const int32_t* charges,
                             const float* bags, const float* leb, uint64_t seed, int32_t mode, void* ws,
                             size_t ws_bytes, float* actions_out, float* out, void* stream) {
  return mg_cov_sample_ids(c, theta, pos, charges, bags, leb, seed, 0, 1, mode, ws, ws_bytes, actions_out, out, stream);
}

extern "C" int mg_canvas_append(int32_t B, int32_t N, int32_t Z, const int32_t* zs_host, const float* actions, double* pos64,
                                float* pos32, int32_t* charges, float* bags, int32_t* natoms, double* newpos,
                                void* stream) {
  if (B < 1 || N < 1 || Z < 2 || Z > MG_MAX_Z) MG_FAIL(MG_EINVAL, "bad canvas shape B=%d N=%d Z=%d", B, N, Z);
  CanvasZs zs;
  for (int i = 0; i < 8; ++i) zs.z[i] = i < Z ? zs_host[i] : 0;
  hipLaunchKernelGGL(k_canvas_append, dim3((B + 127) / 128), dim3(128), 0, (hipStream_t)stream, B, N, Z, zs, actions, pos64,
                     pos32, charges, bags, natoms, newpos);
  LAUNCH_CHECK();
  return MG_OK;
}

// The list-build kernels flag inconsistent inputs in the workspace (encoder.inc: 1 = real atoms not compacted to the
// front of a canvas, 2 = cfg.TA / cfg.TE differ from what `charges` holds).  Reading the flags costs a stream
// synchronisation, so it is a separate call: the Python agent makes it wherever it synchronises anyway.
extern "C" int mg_cov_check(const mg_cov_cfg* c, const void* ws, size_t ws_bytes, void* stream) {
  PLayout P;
  int rc = build_layout(c, &P);
  if (rc) return rc;
  WS w;
  rc = ws_build(c, P, const_cast<void*>(ws), &w, nullptr);
  if (rc) return rc;
  if (ws_bytes < w.bytes) MG_FAIL(MG_ENOMEM, "workspace %zu bytes < required %zu", ws_bytes, w.bytes);
  int flag = 0;
  HIP_CHECK(hipMemcpyAsync(&flag, w.L.err, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
  if (flag == 1) MG_FAIL(MG_EINVAL, "charges: the real atoms of a canvas must be compacted to the front (a padding slot precedes an atom)");
  if (flag == 2) MG_FAIL(MG_EINVAL, "cfg.TA / cfg.TE do not match the atom counts in charges");
  if (flag != 0) MG_FAIL(MG_EINVAL, "list build reported error %d", flag);
  return MG_OK;
}

extern "C" int mg_cov_head_outputs(const mg_cov_cfg* c, const void* ws, size_t ws_bytes, float* out, void* stream) {
  PLayout P;
  int rc = build_layout(c, &P);
  if (rc) return rc;
  WS w;
  rc = ws_build(c, P, const_cast<void*>(ws), &w, nullptr);
  if (rc) return rc;
  if (ws_bytes < w.bytes) MG_FAIL(MG_ENOMEM, "workspace %zu bytes < required %zu", ws_bytes, w.bytes);
  HeadOutSrc src = {w.L.natoms, w.L.atom_off, w.logitF, w.logitE, w.dout, w.logz, {w.cond[0], w.cond[1], w.cond[2], w.cond[3], w.cond[4]}};
  hipLaunchKernelGGL(k_head_outputs, dim3(c->B), dim3(256), 0, (hipStream_t)stream, c->B, c->N, c->Z, c->G, src, out);
  LAUNCH_CHECK();
  return MG_OK;
}

extern "C" int mg_so3_density(int32_t B, int64_t S, int32_t Bp, const float* coef, const float* points,
                              int32_t has_beta, float beta, const float* logz, const uint8_t* empty, int32_t mode,
                              float* out, void* stream) {
  if (B < 1 || S < 0 || (Bp != B && Bp != 1)) MG_FAIL(MG_EINVAL, "points must be [S][B][3] or [S][1][3] (B=%d, Bp=%d)", B, Bp);
  if (mode < 0 || mode > 2 || (mode == 2 && !has_beta)) MG_FAIL(MG_EINVAL, "mode %d not available", mode);
  if (has_beta && !logz) MG_FAIL(MG_EINVAL, "logz is required for the exponential family");
  if (S == 0) return MG_OK;
  const long total = (long)S * B;
  hipLaunchKernelGGL(k_so3_density, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, B, (long)S,
                     Bp, coef, points, has_beta, beta, logz, empty, mode, out);
  LAUNCH_CHECK();
  return MG_OK;
}

// ---- one PPO mini-batch: forward -> float64 loss -> backward in ONE call (and, when asked, ONE graph launch) -----------------
// molgym/ppo.py:124-131 (compute_loss + loss.backward()) for the device-resident path of ppo.train.
// graph_slot < 0: the ~27 launches go to `stream` one by one (what mg_cov_forward + mg_ppo_loss + mg_cov_backward do).
// graph_slot >= 0: the same launch code RECORDS instead (launch.inc); the recording updates the kernel nodes of this host
// thread's cached hipGraphExec number `graph_slot` in place (grids and arguments change with every mini-batch's ragged sizes)
// and one hipGraphLaunch replaces the launches.  Mini-batches in flight on different streams use different slots.  Anything
// the graph form cannot express (side streams of the large configurations, asynchronous memsets) makes the call fall back to
// the eager form by itself; *used_graph reports which one ran.
extern "C" int mg_cov_ppo_step(const mg_cov_cfg* c, const float* theta, const float* pos, const int32_t* charges, const float* bags,
                               const float* actions, const float* leb, void* ws, size_t ws_bytes, const double* old_logp,
                               const double* adv, const double* ret, double clip_ratio, double vf_coef, double entropy_coef,
                               double loss_scale, float* out, float* gout, double* stats, double* stats_accum,
                               float* grad_theta, int32_t graph_slot, int32_t flags, int32_t* used_graph, void* stream) {
  if (!c || !out || !gout || !stats || !grad_theta) MG_FAIL(MG_EINVAL, "mg_cov_ppo_step: null argument");
  if (flags & ~(MG_STEP_WEIGHTS_CURRENT | MG_STEP_DEFER_FOLD)) MG_FAIL(MG_EINVAL, "mg_cov_ppo_step: unknown flags %d", flags);
  hipStream_t s = (hipStream_t)stream;
  static int fuse_loss = -1;
  if (fuse_loss < 0) { const char* e = getenv("MG_FUSED_LOSS"); fuse_loss = e ? atoi(e) : 1; }
  auto run = [&]() -> int {
    // MG_FUSED_LOSS: 1 (default) = the sample's own coefficients in k_heads_fwd, the statistics as a rider of k_heads_bwd [r6];
    // 2 = both in the last workgroup of k_heads_fwd (round 4 / 5); 0 = the loss as its own launch
    const PpoLossArgs la = {old_logp, adv, ret, clip_ratio, vf_coef, entropy_coef, loss_scale, stats, gout, stats_accum,
                            fuse_loss == 1 ? 1 : 0};
    bool fused = false;
    int rc = cov_forward_impl(c, theta, pos, charges, bags, const_cast<float*>(actions), leb, ws, ws_bytes, out, stream, nullptr,
                              fuse_loss ? &la : nullptr, &fused, flags);
    if (rc) return rc;
    if (!fused) {  // (staged heads: the loss as its own launch)
      hipLaunchKernelGGL(k_ppo_loss, dim3(1), dim3(256), 0, s, (int)c->B, (const float*)out, old_logp, adv, ret, clip_ratio, vf_coef,
                         entropy_coef, stats, gout, loss_scale, stats_accum);
      LAUNCH_CHECK();
    }
    const bool rider = fused && la.defer_stats;
    return cov_backward_impl(c, theta, pos, charges, bags, actions, leb, ws, ws_bytes, gout, grad_theta, stream,
                             (flags & MG_STEP_DEFER_FOLD) != 0, rider ? &la : nullptr, rider ? out : nullptr);
  };
  if (used_graph) *used_graph = 0;
  static int graphs_on = -1;
  if (graphs_on < 0) { const char* e = getenv("MG_GRAPH"); graphs_on = e ? atoi(e) : 1; }
  const bool want_graph = graph_slot >= 0 && graph_slot < MG_GRAPH_SLOTS && graphs_on && !g_prof_on &&
                          c->TE < MG_SIDE_MIN_EDGES;  // one stream only: the side-stream forks of the large configurations are events
  if (want_graph) {
    int rc = ensure_tables();  // (its one-time uploads synchronise: not inside a recording)
    if (rc) return rc;
    g_rec.begin(s);
    rc = run();
    g_rec.end();
    if (rc) return rc;
    if (!g_rec.unsupported && !g_rec.recs.empty()) {
      hipError_t e = mg_graph_launch_recorded(g_step_graph[cur_device()][0][graph_slot + ((flags & MG_STEP_WEIGHTS_CURRENT) ? MG_GRAPH_SLOTS : 0)], s);
      if (e == hipSuccess) {
        if (used_graph) *used_graph = 1;
        return MG_OK;
      }
      (void)hipGetLastError();
      static bool warned = false;
      if (!warned) { fprintf(stderr, "molgym_hip: graph launch failed (%s); running eagerly\n", hipGetErrorString(e)); warned = true; }
    }
  }
  return run();
}

// the per-epoch half of the backward: fold the expanded complex weight gradients a workspace accumulated (MG_STEP_DEFER_FOLD)
extern "C" int mg_cov_fold_grads(const mg_cov_cfg* c, void* ws, size_t ws_bytes, float* grad_theta, void* stream) {
  if (!c || !ws || !grad_theta) MG_FAIL(MG_EINVAL, "mg_cov_fold_grads: null argument");
  PLayout P;
  int rc = build_layout(c, &P);
  if (rc) return rc;
  WS w;
  rc = ws_build(c, P, ws, &w, nullptr);
  if (rc) return rc;
  if (ws_bytes < w.bytes) MG_FAIL(MG_ENOMEM, "workspace %zu bytes < required %zu", ws_bytes, w.bytes);
  rc = check_device_of(grad_theta, "grad_theta");
  if (rc) return rc;
  return launch_fold_weights((hipStream_t)stream, w, grad_theta);
}

// the same for the internal-coordinate (SchNet) actor-critic: mg_int_forward + mg_ppo_loss + mg_int_backward in one call and,
// with graph_slot >= 0, one updated graph launch (its ~58 launches are what bounds its update loop on the host)
extern "C" int mg_int_ppo_step(const mg_int_cfg* c, const float* theta, const int32_t* mol_off, const int32_t* edge_off,
                               const int32_t* molZ, const float* molpos, const float* bags, const float* actions, void* ws,
                               size_t ws_bytes, const double* old_logp, const double* adv, const double* ret, double clip_ratio,
                               double vf_coef, double entropy_coef, double loss_scale, float* out, float* gout, double* stats,
                               double* stats_accum, float* grad_theta, int32_t graph_slot, int32_t flags, int32_t* used_graph,
                               void* stream) {
  if (!c || !out || !gout || !stats || !grad_theta) MG_FAIL(MG_EINVAL, "mg_int_ppo_step: null argument");
  if (flags & ~MG_STEP_WEIGHTS_CURRENT) MG_FAIL(MG_EINVAL, "mg_int_ppo_step: unknown flags %d", flags);
  hipStream_t s = (hipStream_t)stream;
  auto run = [&]() -> int {
    static int fuse_loss = -1;
    if (fuse_loss < 0) { const char* e = getenv("MG_FUSED_LOSS"); fuse_loss = e ? atoi(e) : 1; }
    // (MG_FUSED_LOSS as in mg_cov_ppo_step: 1 = coefficients per sample in k_int_heads_eval, statistics as a rider of its adjoint)
    const PpoLossArgs la = {old_logp, adv, ret, clip_ratio, vf_coef, entropy_coef, loss_scale, stats, gout, stats_accum,
                            fuse_loss == 1 ? 1 : 0};
    int rc = int_forward_impl(c, theta, mol_off, edge_off, molZ, molpos, bags, actions, ws, ws_bytes, out, stream, fuse_loss ? &la : nullptr, flags);
    if (rc) return rc;
    if (!fuse_loss) {
      hipLaunchKernelGGL(k_ppo_loss, dim3(1), dim3(256), 0, s, (int)c->B, (const float*)out, old_logp, adv, ret, clip_ratio, vf_coef,
                         entropy_coef, stats, gout, loss_scale, stats_accum);
      LAUNCH_CHECK();
    }
    const bool rider = fuse_loss == 1;
    return int_backward_impl(c, theta, mol_off, edge_off, molZ, molpos, bags, actions, ws, ws_bytes, gout, grad_theta, stream,
                             rider ? &la : nullptr, rider ? out : nullptr);
  };
  if (used_graph) *used_graph = 0;
  static int graphs_on = -1;
  if (graphs_on < 0) { const char* e = getenv("MG_GRAPH"); graphs_on = e ? atoi(e) : 1; }
  if (graph_slot >= 0 && graph_slot < MG_GRAPH_SLOTS && graphs_on && !g_prof_on && c->ME < MG_SIDE_MIN_EDGES) {
    g_rec.begin(s);
    int rc = run();
    g_rec.end();
    if (rc) return rc;
    if (!g_rec.unsupported && !g_rec.recs.empty()) {
      hipError_t e = mg_graph_launch_recorded(g_step_graph[cur_device()][1][graph_slot + ((flags & MG_STEP_WEIGHTS_CURRENT) ? MG_GRAPH_SLOTS : 0)], s);
      if (e == hipSuccess) {
        if (used_graph) *used_graph = 1;
        return MG_OK;
      }
      (void)hipGetLastError();
    }
  }
  return run();
}

// kernel launches the last RECORDED mg_cov_ppo_step of this host thread consisted of (0 if none was recorded): bench.py prints it
extern "C" int mg_cov_step_launches(void) { return (int)g_rec.recs.size(); }

#ifdef MG_TS
// debug builds only (tools/ts_heads.sh): read the phase timestamps (100 MHz ticks) and choose the stamped workgroup
extern "C" int mg_debug_ts(unsigned long long* out, int block) {
  HIP_CHECK(hipDeviceSynchronize());
  HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ts), sizeof(unsigned long long) * 128));
  HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(g_ts_block), &block, sizeof(int)));
  return MG_OK;
}
// (tools/ts_heads_wg.sh) the start / end stamps of every workgroup of the two heads kernels
extern "C" int mg_debug_wg_ts(unsigned long long* out) {
  HIP_CHECK(hipDeviceSynchronize());
  HIP_CHECK(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wg_ts), sizeof(unsigned long long) * 3 * 256 * 4));
  return MG_OK;
}
#endif
