p='molgym_amd/csrc/molgym_hip.hip'
s=open(p).read()
# 1. split arg construction out of prep_weights
old=s[s.index('static int prep_weights(hipStream_t s, const float* theta, WS& w, bool zero_scratch = false, const ListsJob* lj = nullptr) {'):s.index('struct SampleCtx {')]
new='''static void prep_lins(WS& w, std::vector<Lin*>& all) {
  for (int k = 0; k < 3; ++k)
    for (int l = 0; l < 5; ++l) { all.push_back(&w.rad[k][l]); all.push_back(&w.edge[k][l]); all.push_back(&w.atom[k][l]); }
  all.push_back(&w.lin_in);
  for (int l = 0; l < 5; ++l) all.push_back(&w.mix[l]);
  for (int m = 0; m < NMLP; ++m) { all.push_back(&w.mlp[m][0]); all.push_back(&w.mlp[m][1]); }
}
static int prep_weights(hipStream_t s, const float* theta, WS& w, bool zero_scratch = false, const ListsJob* lj = nullptr) {
  std::vector<Lin*> all;
  prep_lins(w, all);
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

'''
s=s.replace(old,new)

# 2. front decision before the prep block
old='''  hipEvent_t weights_ready = nullptr;
  bool lists_done = false;
  {
    hipStream_t ss = side_fork(s);'''
new='''  hipEvent_t weights_ready = nullptr;
  bool lists_done = false;
  // the FRONT form of the fused level-0 kernel (level0.inc) is the step's first launch and does the list build and the weight
  // preparation as roles of its own grid: nothing is launched here then
  const bool front = level0_front(B <= MG_LISTS_SMALL_B && N <= 16, c, w);
  if (front) lists_done = true;
  else {
    hipStream_t ss = side_fork(s);'''
assert old in s
s=s.replace(old,new)

# 3. launch
old='''    ProfScope prof(s, "k_level0");
    hipLaunchKernelGGL(k_level0_fwd, dim3(TA), dim3(L0_T), 0, s, la, w.L);
    LAUNCH_CHECK();
    input_done = true;'''
new='''    ProfScope prof(s, "k_level0");
    WPrepArgs pa;
    memset(&pa, 0, sizeof(pa));
    if (front) {
      std::vector<Lin*> all;
      prep_lins(w, all);
      if (all.size() > WPREP_MAX) MG_FAIL(MG_EINVAL, "level-0 kernel: %zu derived matrices", all.size());
      for (size_t i = 0; i < all.size(); ++i) {
        Lin* Lp = all[i];
        pa.w[i] = {theta + Lp->w_off, Lp->mf, Lp->mb, Lp->O, Lp->Q, Lp->ldf, Lp->ldb, Lp->cplx, Lp->perm_n};
      }
      pa.zero_f = w.dwexp_all; pa.zero_n = w.dwexp_floats; pa.zero_i4 = nullptr;  // (the list workgroup clears the error flags)
      la.in_w_off = (int)P.in_w;
      bool al = ((uintptr_t)theta & 15) == 0 && P.rad_scales[0] % 4 == 0 && (P.rad_scales[1] - P.rad_scales[0]) % 4 == 0;
      for (int l = 0; l < 5; ++l) {
        la.edge_w_off[l] = (int)P.edge_w[0][l]; la.edge_K[l] = 2 * P.edge_cin[0][l];
        la.atom_w_off[l] = (int)P.atom_w[0][l]; la.atom_K[l] = 2 * P.atom_tau[0][l];
        al = al && P.edge_w[0][l] % 4 == 0 && P.atom_w[0][l] % 4 == 0;
      }
      la.theta16 = al ? 1 : 0;
      la.B = B; la.cfgTA = TA; la.cfgTE = TE; la.n_prep = (int)all.size();
      hipLaunchKernelGGL((k_level0_fwd<true>), dim3(TA + 1 + 4 * la.n_prep), dim3(L0_T), 0, s, la, w.L, pa);
    } else {
      hipLaunchKernelGGL((k_level0_fwd<false>), dim3(TA), dim3(L0_T), 0, s, la, w.L, pa);
    }
    LAUNCH_CHECK();
    input_done = true;'''
assert old in s
s=s.replace(old,new)
# fused0 when front must be true: level0_fused(lists_done...) unchanged (lists_done true)
open(p,'w').write(s)

p='molgym_amd/csrc/level0.inc'
s=open(p).read()
s=s.replace('''struct L0Args {''','''// the FRONT form (the step's first launch does the lists and the weight preparation itself): MG_FRONT=0 keeps them a launch of
// their own in front of k_level0_fwd<false>
static bool level0_front(bool lists_small, const mg_cov_cfg* c, const WS& w) {
  static int on = -1;
  if (on < 0) { const char* e = getenv("MG_FRONT"); on = e ? atoi(e) : 1; }
  return on && level0_fused(lists_small, c, w) && !side_active();
}

struct L0Args {''',1)
open(p,'w').write(s)
