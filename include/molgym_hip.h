/* molgym_hip.h -- C ABI of the MI355X (gfx950) PPO policy/value hot path.
 *
 * Boundary replaced (reference = gncs/molgym, all paths under /root/reference):
 *   mg_cov_forward      <- CovariantAC.step(observations, actions) with actions given,
 *                          molgym/agents/covariant/agent.py:209-334 (called from
 *                          molgym/ppo.py:26), after parse_observations (:165-197) has
 *                          been done on the host into the padded arrays taken here.
 *   mg_cov_backward     <- autograd of loss.backward(), molgym/ppo.py:131, through the
 *                          same step(); gradients ACCUMULATE into grad_theta exactly as
 *                          .grad does over mini-batches (ppo.py:122-131).
 *   mg_ppo_loss         <- compute_loss arithmetic, molgym/ppo.py:28-52 (float64 seam).
 *   mg_gae              <- DynamicPPOBuffer.finish_path, molgym/buffer.py:74-82 +
 *                          util.discount_cumsum, molgym/tools/util.py:72-87.
 *   mg_adv_normalize    <- DynamicPPOBuffer.get_data, molgym/buffer.py:104-110.
 *   mg_grad_norm_clip   <- util.compute_gradient_norm (util.py:61-69) +
 *                          torch.nn.utils.clip_grad_norm_ (ppo.py:144).
 *
 * All pointers are DEVICE pointers unless a name ends in _host.  No ownership is taken.
 * Every entry point returns 0 on success, a negative MG_E* code otherwise, and never
 * throws; mg_last_error() gives the text.  Launches go to `stream` (a hipStream_t passed
 * as void*), nothing synchronises the device.
 */
#ifndef MOLGYM_HIP_H
#define MOLGYM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MG_OK 0
#define MG_EINVAL (-1)   /* unsupported / inconsistent configuration */
#define MG_EHIP (-2)     /* HIP runtime error */
#define MG_ENOMEM (-3)   /* workspace too small */

#define MG_MAX_Z 8

/* Fixed by the build (reference defaults, molgym/tools/arg_parser.py:55-60):
 * maxl = 4, num_cg_levels = 3, num_channels_hidden = 10, num_channels_per_element = 4; the last three are -D parameters
 * of other builds of the same sources (mg_cov_build_params). */
typedef struct mg_cov_cfg {
  int32_t B;            /* samples in the mini-batch                                   */
  int32_t N;            /* canvas_size                                                 */
  int32_t Z;            /* len(zs), zs[0] == 0 is the null symbol                      */
  int32_t zs[MG_MAX_Z]; /* atomic numbers                                              */
  int32_t W;            /* network_width (multiple of 4)                               */
  int32_t G;            /* num_gaussians                                               */
  int32_t TA;           /* total real atoms in the batch  = sum_b n_b                  */
  int32_t TE;           /* total real edges in the batch  = sum_b n_b^2                */
  int32_t has_beta;     /* 1: ExpSO3Distribution(beta); 0: SO3Distribution             */
  float beta;
  float bag_scale;
  float min_distance, max_distance;
} mg_cov_cfg;

const char* mg_last_error(void);
/* MG_ABI_VERSION is bumped whenever an entry point is added / changed or the workspace layout changes; the binding
 * (molgym_amd/_lib.py::_bind) refuses a library whose mg_abi_version() differs, so a stale prebuilt .so is caught by the
 * version and not by a missing symbol.  1: rounds 1-2; 2: mg_cov_channels, mg_cov_sample_ids, channel-major workspace; 3: mg_cov_ppo_step; 4: mg_ppo_epoch_end, mg_adam_step_gated; 5: mg_cov_step_launches; 6: mg_int_ppo_step; 7: mg_cov_build_params (num_cg_levels a build parameter); 8: mg_cov_ppo_step takes `flags`, mg_cov_fold_grads, derived weights first in the workspace; 9: mg_int_ppo_step takes `flags` (MG_STEP_WEIGHTS_CURRENT), derived weights first in the SchNetAC workspace too.  */
#define MG_ABI_VERSION 9
int mg_abi_version(void);
/* num_channels_hidden / num_channels_per_element THIS build of the library was compiled for (tools/arg_parser.py:55-60;
 * covariant/agent.py:64,82-83 derive every SO3Tau from them): compile-time constants of the kernels, 10 / 4 by default.
 * Other values are other builds of the same sources (hipcc -DCH=.. -DCE=..); molgym_amd/_lib.py builds / loads them.   */
int mg_cov_channels(int32_t* hidden, int32_t* per_element);
/* the same, with maxl and num_cg_levels (arg_parser.py:55-56).  num_cg_levels is a build parameter too (hipcc -DNLEV=2..4:
 * the level loops, arena and parameter layout follow it; the one-launch-per-level kernels of the small mini-batches are
 * written for 3 and fall back to the general launches otherwise); maxl = 4 is fixed (tables, thread maps, LDS layouts).   */
int mg_cov_build_params(int32_t* hidden, int32_t* per_element, int32_t* maxl, int32_t* num_cg_levels);

/* ---- optional kernel-span timing (measurement only) ------------------------------ */
/* on != 0: forward/backward bracket their dominant kernels with HIP events recorded on
 * the launch stream; mg_profile_report writes "name total_ms count" lines (host buffer). */
int mg_profile_enable(int on);
int mg_profile_report(char* buf_host, size_t cap);

/* ---- parameter vector ------------------------------------------------------------- */
/* theta is ONE flat float32 vector.  Slot order and shapes: molgym_amd/layout.py.       */
int mg_cov_num_params(const mg_cov_cfg* cfg, int64_t* num_params);
/* offsets_out[i] = float offset of slot i (n_slots entries, plus the total at the end). */
int mg_cov_param_offsets(const mg_cov_cfg* cfg, int64_t* offsets_out_host, int32_t* n_slots_host);

/* ---- workspace -------------------------------------------------------------------- */
int mg_cov_workspace_bytes(const mg_cov_cfg* cfg, size_t* bytes_host);
/* Look up a named intermediate inside the workspace (tests only): float offset + count. */
int mg_cov_workspace_lookup(const mg_cov_cfg* cfg, const char* name, int64_t* offset_floats_host,
                            int64_t* count_floats_host);

/* ---- forward / backward of CovariantAC.step(obs, actions) ------------------------- */
/* pos      [B][N][3] f32, real atoms first, zero padded (covariant/tools.py:18-31)
 * charges  [B][N] i32 atomic numbers, 0 = padding
 * bags     [B][Z] f32
 * actions  [B][6] f32: focus, element index, distance, ox, oy, oz (agent.py:148-154)
 * leb      [51][1730] f32, feature-major: rows 2q / 2q+1 = Re / Im of Y_q ('qm', q = l*l+l+m) at the
 *          Lebedev-71 points, row 50 = log weight (weights sum to 1); molgym_amd/lebedev.py
 * out      [3][B] f32: logp, ent, v
 * ws       workspace of mg_cov_workspace_bytes(); keeps what backward needs.          */
int mg_cov_forward(const mg_cov_cfg* cfg, const float* theta, const float* pos, const int32_t* charges,
                   const float* bags, const float* actions, const float* leb, void* ws, size_t ws_bytes,
                   float* out, void* stream);
/* Rollout-side step(obs) (actions = None, agent.py:229-292): the four sub-actions are DRAWN on the device as
 * the heads run -- mode 1 = self.training (Categorical / GMM / rejection samplers), mode 2 = evaluation
 * (argmax variants) -- written to actions_out [B][6], and logp / ent / v of the drawn actions to out [3][B].
 * Counter-based RNG keyed by `seed` (the torch RNG stream itself is not reproducible here).               */
int mg_cov_sample(const mg_cov_cfg* cfg, const float* theta, const float* pos, const int32_t* charges,
                  const float* bags, const float* leb, uint64_t seed, int32_t mode, void* ws, size_t ws_bytes,
                  float* actions_out, float* out, void* stream);
/* The same with the random stream of row b keyed by sample_base + sample_stride * b instead of b: a rollout stepped in
 * GROUPS of environments (molgym_amd/ppo.py::_rollout_pipelined overlaps one group's host-side reward,
 * reward.py:36-55, with the other group's policy evaluation; env_container.py:11-74 step_async / step_wait) then draws for
 * every environment what ONE launch over all of them draws.  mg_cov_sample == base 0, stride 1.          */
int mg_cov_sample_ids(const mg_cov_cfg* cfg, const float* theta, const float* pos, const int32_t* charges,
                      const float* bags, const float* leb, uint64_t seed, int32_t sample_base, int32_t sample_stride,
                      int32_t mode, void* ws, size_t ws_bytes, float* actions_out, float* out, void* stream);
/* gout [3][B] f32: dL/dlogp, dL/dent, dL/dv.  grad_theta += dL/dtheta.                 */
int mg_cov_backward(const mg_cov_cfg* cfg, const float* theta, const float* pos, const int32_t* charges,
                    const float* bags, const float* actions, const float* leb, void* ws, size_t ws_bytes,
                    const float* gout, float* grad_theta, void* stream);

/* ---- persistent device canvases of the rollout (replaces the per-step re-parse of every observation,
 * covariant/agent.py:165-197 + covariant/tools.py:8-49, between two steps of an episode) -----------------------------
 * Appends the atom an action row places (to_action_space, agent.py:147-163) to each canvas IN PLACE:
 *   new position = pos64[focus] + (double)(float)(distance * direction)   (origin on an empty canvas),
 *   pos64 / pos32 / charges slot natoms <- it, bags[element] -= 1, natoms += 1   (skipped for a null element or a full canvas)
 * actions [B][6] f32 (mg_cov_sample's actions_out); pos64 [B][N][3] f64 (the environment's own precision), pos32 its f32
 * mirror (what mg_cov_sample / mg_cov_forward read), charges [B][N] i32, bags [B][Z] f32, natoms [B] i32;
 * newpos [B][3] f64 out: the positions placed (what the host hands to the environments); zs_host: HOST array of Z ints. */
int mg_canvas_append(int32_t B, int32_t N, int32_t Z, const int32_t* zs_host, const float* actions, double* pos64,
                     float* pos32, int32_t* charges, float* bags, int32_t* natoms, double* newpos, void* stream);

/* ---- mini-batch gather (replaces collect_data_batch, molgym/ppo.py:77-81, on a rollout parked in HBM) ----------------
 * dst[f][b][:] = src[f][idx[b]][:] for nf <= 8 row-major matrices in ONE launch; row_bytes[f] (multiples of 4) are HOST
 * values, src / dst are HOST arrays of device pointers, idx [B] int64 on the device.  ppo.train draws a new permutation
 * every epoch, so every mini-batch is a gather of positions, charges, bags, action rows and the three float64 columns of
 * the loss: seven index_select calls cost more host time than the forward pass's launches.                            */
int mg_gather_rows(int32_t nf, const void* const* src, void* const* dst, const int32_t* row_bytes, const int64_t* idx,
                   int32_t B, void* stream);

/* Input validation of the forward that just ran on `ws`: synchronises `stream`, then MG_EINVAL if the list build found
 * real atoms that are not compacted to the front of their canvas, or cfg.TA / cfg.TE inconsistent with `charges`
 * (the forward itself never synchronises, so it cannot report these).                                              */
int mg_cov_check(const mg_cov_cfg* cfg, const void* ws, size_t ws_bytes, void* stream);

/* ---- head outputs of the forward that just ran (what step()'s `dists` are built from, agent.py:224-286,325-331) ----
 * Packs, sample-major, out of the workspace:  focus logits [B][N] (0 beyond the sample's atoms) | natoms [B] (as f32) |
 * element logits [B][Z] | GMM head [B][2G] (mixture logits, then pre-tanh means) | conditioned orientation
 * coefficients [B][25][CE = 4][2] (q = l*l+l+m) | log Z [B].   out holds B * (N + 1 + Z + 2G + 200 + 1) floats.   */
int mg_cov_head_outputs(const mg_cov_cfg* cfg, const void* ws, size_t ws_bytes, float* out, void* stream);

/* ---- orientation density on caller-supplied points ------------------------------------------------------
 * What dists[-1] of CovariantAC.step()'s return (covariant/agent.py:325-331) evaluates in .log_prob(value) /
 * .prob(value): ExpSO3Distribution (spherical_dists.py:273-286) when has_beta, SO3Distribution (:160-179) otherwise.
 *   coef   [B][25][2] f32 channel-summed coefficients AFTER normalize_alms (so3_tools.py:68-75), q = l*l+l+m
 *   points [S][Bp][3] f32, Bp == B, or 1 to broadcast one grid over the samples; normalised to unit length
 *          inside (cormorant SphericalHarmonics(normalize=True)), a zero vector keeps Y_00 only
 *   logz   [B] f32 (has_beta), empty [B] u8 or NULL (SO3Distribution's uniform density on an empty canvas)
 *   mode   0 = log_prob, 1 = prob, 2 = unnormalised log_prob (has_beta only);  out [S][B] f32               */
int mg_so3_density(int32_t B, int64_t S, int32_t Bp, const float* coef, const float* points, int32_t has_beta,
                   float beta, const float* logz, const uint8_t* empty, int32_t mode, float* out, void* stream);

/* ---- internal-coordinate agent: SchNetAC.step(obs, actions), molgym/agents/internal/agent.py:181-353 ----
 * The schnetpack SchNet embedding is evaluated once over 3B molecules: set 0 = the canvases (n_b atoms), sets 1
 * and 2 = canvas + the new atom placed by zmat.position_atom_helper (internal/zmat.py:96-133, done on the host
 * in float64) at +dihedral / -dihedral.  Molecule m = set * B + b.
 *   mol_off  [3B+1] i32 atom offsets, edge_off [3B+1] i32 ordered-pair offsets (n(n-1) per molecule),
 *   molZ [MA] i32 atomic numbers, molpos [MA][3] f32, bags [B][Z] f32,
 *   actions [B][7] f32: stop, focus, element, distance, angle, dihedral, kappa (agent.py:26,306-308)
 *   out [3][B] f32: logp (masked sum), ent (focus + element), v.                                            */
typedef struct mg_int_cfg {
  int32_t B, N, Z;
  int32_t zs[MG_MAX_Z];
  int32_t W;          /* network_width, multiple of 16                                 */
  int32_t TA;         /* real atoms on the canvases                                    */
  int32_t MA;         /* atoms over all 3B molecules = 3*TA + 2*B                      */
  int32_t ME;         /* ordered pairs over all 3B molecules                           */
  float min_distance, max_distance;
} mg_int_cfg;
int mg_int_num_params(const mg_int_cfg* cfg, int64_t* num_params);
int mg_int_param_offsets(const mg_int_cfg* cfg, int64_t* offsets_out_host, int32_t* n_slots_host);
int mg_int_workspace_bytes(const mg_int_cfg* cfg, size_t* bytes_host);
/* Float offset / count of a named intermediate inside the workspace ("logitF" [TA], "logitE" [B][Z], "cout" [B][3]
 * = phi_continuous output before tanh, "kv" [2][B] kappa logits, ...).  The rollout-side step(obs) of the Python
 * agent reads the distribution parameters of each sub-action through it (internal/agent.py:206-292). */
int mg_int_workspace_lookup(const mg_int_cfg* cfg, const char* name, int64_t* offset_floats_host,
                            int64_t* count_floats_host);
int mg_int_forward(const mg_int_cfg* cfg, const float* theta, const int32_t* mol_off, const int32_t* edge_off,
                   const int32_t* molZ, const float* molpos, const float* bags, const float* actions, void* ws,
                   size_t ws_bytes, float* out, void* stream);
int mg_int_backward(const mg_int_cfg* cfg, const float* theta, const int32_t* mol_off, const int32_t* edge_off,
                    const int32_t* molZ, const float* molpos, const float* bags, const float* actions, void* ws,
                    size_t ws_bytes, const float* gout, float* grad_theta, void* stream);

/* ---- PPO loss, float64 (ppo.py:28-52) --------------------------------------------- */
/* pred [3][B] f32 (logp, ent, v); old_logp, adv, ret [B] f64.
 * stats[6] f64: policy_loss, entropy_loss, vf_loss, total_loss, approx_kl, clip_fraction.
 * gout [3][B] f32 = d total_loss / d pred (may be NULL).                                */
int mg_ppo_loss(int32_t B, const float* pred, const double* old_logp, const double* adv, const double* ret,
                double clip_ratio, double vf_coef, double entropy_coef, double* stats, float* gout,
                void* stream);

/* ---- one PPO mini-batch in one call: step(obs, actions) -> loss -> backward (molgym/ppo.py:124-131) ---------------------
 * What the device-resident update loop of ppo.train does per mini-batch: mg_cov_forward, mg_ppo_loss, mg_cov_backward on the
 * same stream, with
 *   out [3][B], gout [3][B] f32 and stats[6] f64 caller-owned (kept: `out` is what step() returned, gout = d loss / d out);
 *   loss_scale multiplies gout (this rank's share of a data-parallel mini-batch; 1 otherwise);
 *   stats_accum[6] f64 (may be NULL): += loss_scale * stats, atomically -- the epoch's mean of mini-batch means;
 *   gradients ACCUMULATE into grad_theta.
 * graph_slot < 0: ~27 stream launches.  graph_slot in [0, 8): the launches are recorded, the kernel nodes of this host
 * thread's cached graph number graph_slot are updated in place (grids and arguments follow the mini-batch's ragged sizes) and
 * ONE hipGraphLaunch is issued -- ~30 us of host time instead of ~250 (the update loop is host-bound otherwise).  Mini-batches
 * in flight on different streams must use different slots.  Falls back to the stream launches by itself where a graph cannot
 * express the step (side streams of the large configurations); *used_graph_host (may be NULL) reports which form ran.
 * MG_GRAPH=0 in the environment disables the graph form.
 * flags (theta is constant over the mini-batches of a PPO epoch -- the reference steps the optimizer once per epoch,
 * ppo.py:117-146 -- so what depends on theta alone need not be redone per mini-batch):
 *   MG_STEP_WEIGHTS_CURRENT  the derived weight matrices in THIS workspace (they sit first in it, at offsets that do not depend on
 *                            the mini-batch) were written by an earlier call with the same theta: skip their preparation;
 *   MG_STEP_DEFER_FOLD       leave the expanded complex weight gradients accumulating in the workspace instead of folding them
 *                            into grad_theta at the end of the step; the caller folds once per epoch with mg_cov_fold_grads
 *                            (before it reads grad_theta) -- one fold per workspace used.                                      */
#define MG_STEP_WEIGHTS_CURRENT 1
#define MG_STEP_DEFER_FOLD 2
int mg_cov_ppo_step(const mg_cov_cfg* cfg, const float* theta, const float* pos, const int32_t* charges, const float* bags,
                    const float* actions, const float* lebedev, void* workspace, size_t workspace_bytes,
                    const double* old_logp, const double* adv, const double* ret, double clip_ratio, double vf_coef,
                    double entropy_coef, double loss_scale, float* out, float* gout, double* stats, double* stats_accum,
                    float* grad_theta, int32_t graph_slot, int32_t flags, int32_t* used_graph_host, void* stream);
/* grad_theta += the expanded complex weight gradients a workspace accumulated over mg_cov_ppo_step(.., MG_STEP_DEFER_FOLD) calls
 * (and the accumulator is left zero): the per-epoch half of what mg_cov_backward does at the end of every call.             */
int mg_cov_fold_grads(const mg_cov_cfg* cfg, void* workspace, size_t workspace_bytes, float* grad_theta, void* stream);

/* the same for the internal-coordinate agent (mg_int_forward + mg_ppo_loss + mg_int_backward; the loss is evaluated by the last
 * workgroup of the forward's last launch).  flags: MG_STEP_WEIGHTS_CURRENT as above (the derived weight matrices sit first in the
 * workspace, at offsets that do not depend on the batch); this agent has no expanded gradients to fold.                       */
int mg_int_ppo_step(const mg_int_cfg* cfg, const float* theta, const int32_t* mol_off, const int32_t* edge_off,
                    const int32_t* molZ, const float* molpos, const float* bags, const float* actions, void* workspace,
                    size_t workspace_bytes, const double* old_logp, const double* adv, const double* ret, double clip_ratio,
                    double vf_coef, double entropy_coef, double loss_scale, float* out, float* gout, double* stats,
                    double* stats_accum, float* grad_theta, int32_t graph_slot, int32_t flags, int32_t* used_graph_host,
                    void* stream);

/* kernel launches (= graph nodes) of the last mg_cov_ppo_step this host thread issued in graph form; 0 if none (measurement) */
int mg_cov_step_launches(void);

/* ---- GAE-lambda over concatenated trajectories (buffer.py:74-82) ------------------ */
/* path_off [P+1] i32 start offsets; rew, val [T] f64; last_val [P] f64.
 * adv[t] = discount_cumsum(delta, gamma*lam), ret[t] = discount_cumsum(rew ++ last_val, gamma)[:-1] */
int mg_gae(int32_t num_paths, const int32_t* path_off, const double* rew, const double* val,
           const double* last_val, double gamma, double lam, double* adv, double* ret, void* stream);
/* adv <- (adv - mean) / std, population std, no epsilon (buffer.py:104-110).            */
int mg_adv_normalize(int32_t T, double* adv, double* scratch2, void* stream);

/* ---- gradient norm + clip (util.py:61-69, ppo.py:144) ------------------------------ */
/* norm_out[0] = ||g||_2 (f32, device; norm_out must hold 2 floats, [1] is scratch).
 * If max_norm > 0: g *= min(1, max_norm/(norm+1e-6)).                                   */
int mg_grad_norm_clip(int64_t n, float* grad, float max_norm, float* norm_out, void* stream);

/* One Adam update of the flat parameter vector (replaces optimizer.step(), molgym/ppo.py:145, for torch.optim.Adam on the
 * single flat theta): the arithmetic of torch.optim.Adam's foreach implementation in one launch --
 *   grad' = (maximize ? -grad : grad) + weight_decay * param;  exp_avg.lerp_(grad', 1 - beta1);
 *   exp_avg_sq = beta2 * exp_avg_sq + (1 - beta2) * grad'^2;  [amsgrad: max_exp_avg_sq = max(max_exp_avg_sq, exp_avg_sq)]
 *   param -= lr / (1 - beta1^step) * exp_avg / (sqrt(exp_avg_sq or its running max) / sqrt(1 - beta2^step) + eps)
 * on the optimizer's OWN state tensors (so optimizer.state_dict() stays what torch would have produced).
 * `step` is the count AFTER this update (>= 1); max_exp_avg_sq may be null (amsgrad off); all arrays [n] f32 on the device. */
int mg_adam_step(int64_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* max_exp_avg_sq,
                 double lr, double beta1, double beta2, double eps, double weight_decay, int64_t step, int32_t maximize,
                 void* stream);
/* the same update, skipped entirely when *skip_flag != 0 (device int; may be null) */
int mg_adam_step_gated(int64_t n, float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* max_exp_avg_sq,
                       double lr, double beta1, double beta2, double eps, double weight_decay, int64_t step, int32_t maximize,
                       const int32_t* skip_flag, void* stream);

/* ---- end of a PPO epoch on the device (molgym/ppo.py:133-146: KL test, gradient norm, clip) --------------------------------
 * stats_accum[6] f64 = sum over the epoch's mini-batches of (share x statistics) (mg_cov_ppo_step); inv_num_minibatches = 1 / M.
 * rec[8] f64 <- {policy_loss, entropy_loss, vf_loss, total_loss, approx_kl, clip_fraction} means, the PRE-clip gradient norm,
 * and 1.0 if the update loop has stopped (this epoch's approx_kl > kl_limit, or an earlier epoch's was): the reference breaks
 * BEFORE the optimizer step in that case (ppo.py:139-141).  *stop_flag (device int, zero before the first epoch) latches; the
 * clip (grad *= min(1, max_norm / (norm + 1e-6)), max_norm > 0) is applied only while it is clear, and mg_adam_step_gated with
 * the same flag does nothing once it is set -- so the host may issue the next epoch without waiting for this one and read
 * `rec` late.  scratch: 1 float.                                                                                          */
int mg_ppo_epoch_end(int64_t n, float* grad, float max_norm, const double* stats_accum, double inv_num_minibatches,
                     double kl_limit, double* rec, int32_t* stop_flag, float* scratch, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MOLGYM_HIP_H */
