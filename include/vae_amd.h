/*
 * vae_amd.h -- C ABI of libvae_amd.so: the MI355X (gfx950) kernels behind the
 * reparameterizer sample + KL + NLL + ELBO hot path of the jramapuram/vae pool.
 *
 * The reference has no FFI: its boundary is a duck-typed Python protocol
 * (SURVEY.md section 8b).  Each entry point below therefore cites the reference
 * op chain (file:line under /root/reference) that it replaces; the Python classes
 * in vae_amd/reparameterizers and vae_amd/elbo.py keep the reference's names and
 * signatures and call these functions through ctypes.
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless stated; no torch types;
 *   - nothing here allocates, frees or synchronises: the caller owns every
 *     buffer, calls are asynchronous on `stream` (a hipStream_t, 0 = default);
 *   - `dtype` selects the tensor element type (VAE_F32 / VAE_BF16 / VAE_F16, the last for the reference's
 *     `half=True` configs); arithmetic and
 *     reductions are always fp32 inside the kernels;
 *   - rows are samples.  `row_stride` / `stride_*` / `*_stride` are in ELEMENTS; tensors
 *     without a stride argument are contiguous; an output / gradient stride of 0 means
 *     contiguous too.  Output strides let ConcatReparameterizer children write straight into
 *     the concatenated sample and into one shared d(logits) buffer (no cat / slice copies);
 *   - `workspace` is scratch of at least vae_workspace_bytes() bytes, ZEROED ONCE by the
 *     caller when it is created (the kernels leave its counters at zero again), used by one
 *     call at a time (per stream); the batch means are finished inside the producing kernel;
 *   - return value: VAE_OK, VAE_ERR_ARG (bad shape / null / dtype) or
 *     VAE_ERR_LAUNCH (hipGetLastError() != hipSuccess after the launches);
 *   - a call with B == 0 is a no-op that returns VAE_OK.
 */
#ifndef VAE_AMD_H_
#define VAE_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* vae_stream_t; /* hipStream_t */

enum { VAE_OK = 0, VAE_ERR_ARG = 1, VAE_ERR_LAUNCH = 2 };
enum { VAE_F32 = 0, VAE_BF16 = 1, VAE_F16 = 2 };
enum { VAE_NLL_BERNOULLI = 0, VAE_NLL_GAUSSIAN = 1, VAE_NLL_DISC_LOGISTIC = 2 };

#define VAE_MAX_ROW (1 << 24) /* longest supported row (elements) */
/* softmax-class kernels keep the row in registers up to 4 x 256 16-byte vectors (K <= 4096 fp32, twice that in
 * the 16-bit types when K % 8 == 0; 1024 for unaligned rows, K % 4 != 0): one read of every input.  Longer rows
 * (up to VAE_MAX_ROW) run one block per row and stream the row three times (maxima, sums, outputs).          */
#define VAE_MAX_SOFTMAX_ROW 4096
#define VAE_MAX_SAMPLES 8   /* decoder outputs one vae_nll_multi_elbo_* call combines */
#define VAE_MAX_STEPS 128   /* time steps one vae_nll_steps_elbo_* call scores (pointer tables in the kernel arguments) */
#define VAE_MAX_BN_ROW 4096 /* BatchNormFlow layers up to this width (1024 unaligned) take their column moments in one
                             * launch; wider ones (any F <= VAE_MAX_ROW) one launch per window of that many columns */

/* ABI version of this header; bumped on any signature change. */
int vae_abi_version(void);
/* Bytes of scratch every entry point that takes `workspace` may use (ABI v11: 68 KiB of counters and block partials
 * + 4 MiB of per-chunk row sums for the long-row one-pass NLL = 4 263 936 bytes). */
size_t vae_workspace_bytes(void);
/* Tuning hook (no reference counterpart; scripts/ab.py): overrides one launch-geometry choice for the calls that
 * follow, process-wide.  name: "G", "U", "MAXGRID", "REDUCEGRID" (the VAE_AMD_<name> environment variables set the same knobs
 * once, at the first launch); value 0 restores the built-in choice.  Results do not depend on any of them beyond
 * the summation order of the batch means.                                                         */
int vae_tuning_set(const char* name, int64_t value);

/* In-kernel Philox draws under HIP graph capture.  Every entry point that draws noise takes (seed, offset); the
 * noise of element (row, col) is Philox4x32-10 keyed by (seed, offset, row0 + row, col), `offset` a call counter the
 * caller advances.  A host value recorded into a HIP graph would repeat the same draw on every replay, so `offset`
 * may instead be  VAE_OFFSET_ON_DEVICE | (address of a uint64 device word)  : the kernels read the counter from
 * that word.  vae_rng_next writes such a word from a per-GPU counter held by the library (VAE_GRAPH_OFFSET_BASE +
 * calls so far + 1: disjoint from host counters) and advances the counter by n (n draws: offset, offset + 1, ...);
 * recorded into the graph in front of the drawing kernel it hands every replay a fresh offset, and the backward
 * kernel, given the same word, regenerates that replay's noise.  The word must stay allocated while the graph
 * lives.  vae_rng_reset zeroes the per-GPU counter (reproducible runs).  No reference counterpart: the reference
 * draws with torch's generator (isotropic_gaussian.py:88, gumbel.py:269).                                    */
#define VAE_OFFSET_ON_DEVICE (1ull << 63)
#define VAE_GRAPH_OFFSET_BASE (1ull << 62)
int vae_rng_next(uint64_t* cell, int64_t n, vae_stream_t stream);
int vae_rng_reset(vae_stream_t stream);

/* "Saw a NaN" word (replaces the per-step host check nan_check_and_break of abstract_vae.py:366-369,
 * isotropic_gaussian.py:89-90, vrnn.py:427-428, which synchronises the stream on every step).  The closing wave of
 * every kernel that finishes batch means ORs VAE_NAN_* bits into one word of `workspace` when a mean comes out NaN
 * (one NaN sample makes its batch sum NaN): no extra traffic, no synchronisation.  The word is sticky.
 * vae_nan_flag copies it to *flags_out (host memory) after the work queued on `stream` so far has finished -- it
 * synchronises `stream`: call it once per epoch / every N steps, not per step -- and, with clear != 0, zeroes it.
 * vae_nll_elbo_fused_static keeps the word of its built-in scratch (pass workspace = NULL here to read that one;
 * the device that owns `stream` must be current).                                                            */
enum { VAE_NAN_NLL = 1,     /* nll / loss / elbo means (the decoder's output or the target held a NaN / inf-inf) */
       VAE_NAN_KLD = 2,     /* kld mean */
       VAE_NAN_LATENT = 4,  /* a reparameterizer's parameter means (mu / logvar, concentrations): NaN in the encoder logits */
       VAE_NAN_OTHER = 8 }; /* additional-loss / mutual-information means, the Flow "KL" mean */
int vae_nan_flag(float* workspace, int clear, uint32_t* flags_out, vae_stream_t stream);

/* ---- IsotropicGaussian ------------------------------------------------------
 * reparameterizers/isotropic_gaussian.py:98-131 (split + eps_tol), :77-96
 * (z = e*(0.5*s) + mu, NO exp; eval: z = mu), :158-170 (KL vs N(0,1)),
 * :204-215 (mu_mean, logvar_mean).
 *   logits [B, 2L] rows `row_stride` apart; noise [B, L] or NULL (NULL + training:
 *   Philox4x32-10 N(0,1) keyed by (seed, offset, row0 + row, col), so the draw is
 *   independent of how rows are sharded); z [B, L] (may be NULL when !training:
 *   the reference then returns the mu view itself); s_out [B, L] or NULL (the
 *   reference's params['gaussian']['logvar'] = logits[:, L:] + eps_tol);
 *   kl [B] in `dtype`; means: 2 floats (sum mu, sum logvar) * mean_scale, or NULL.
 *   mean_scale is 1/(B*L) on one GPU and 1/(B_global*L) when the batch is
 *   sharded (the caller then all-reduces the two floats).                        */
int vae_gauss_reparam_kl_fwd(const void* logits, int64_t row_stride, const void* noise, void* z, int64_t z_stride,
                             void* s_out, void* kl, float* means, float* workspace, int64_t B, int64_t L, float eps_tol,
                             float mean_scale, int training, uint64_t seed, uint64_t offset, int64_t row0,
                             int dtype, vae_stream_t stream);
/* Backward of the above: glogits [B, 2L] contiguous from gz [B, L], gs (grad of
 * the logvar output) [B, L], gkl [B], gmeans (2 floats); each grad may be NULL. */
int vae_gauss_reparam_kl_bwd(const void* logits, int64_t row_stride, const void* noise, const void* gz,
                             int64_t gz_stride, const void* gs, const void* gkl, const float* gmeans, void* glogits,
                             int64_t glogits_stride, int64_t B,
                             int64_t L, float eps_tol, float mean_scale, int training, uint64_t seed,
                             uint64_t offset, int64_t row0, int dtype, vae_stream_t stream);

/* The same two for 4-D logits [B, 2C, H, W] (contiguous), which the reference splits on the CHANNEL dimension
 * (isotropic_gaussian.py:108-114: mu = logits[:, :C], s = logits[:, C:] + eps) while the KL still sums over the last
 * dimension only (:170): rows are (b, c, h), W elements each.  noise / z / s_out / gz / gs: [B, C, H, W] contiguous;
 * kl / gkl: [B, C, H]; glogits: [B, 2C, H, W]; means as above with mean_scale = 1 / (B*C*H*W).  The logits are read
 * where they are: no repacked [mu(W) | s(W)] copy.                                                              */
int vae_gauss_reparam_kl_fwd_nchw(const void* logits, const void* noise, void* z, void* s_out, void* kl, float* means,
                                  float* workspace, int64_t B, int64_t C, int64_t H, int64_t W, float eps_tol,
                                  float mean_scale, int training, uint64_t seed, uint64_t offset, int64_t row0,
                                  int dtype, vae_stream_t stream);
int vae_gauss_reparam_kl_bwd_nchw(const void* logits, const void* noise, const void* gz, const void* gs, const void* gkl,
                                  const float* gmeans, void* glogits, int64_t B, int64_t C, int64_t H, int64_t W,
                                  float eps_tol, float mean_scale, int training, uint64_t seed, uint64_t offset,
                                  int64_t row0, int dtype, vae_stream_t stream);

/* KL(N(mu_q, s_q) || N(mu_p, s_p)) summed over the last dim; mu_p = s_p = NULL
 * means N(0,1).  reparameterizers/isotropic_gaussian.py:172-190 with
 * torch/distributions/kl.py:468-471 (scale is used as given: no exp).
 * All four inputs are strided row views [B, L]; kl [B].                         */
int vae_gauss_kl_fwd(const void* mu_q, int64_t stride_mu_q, const void* s_q, int64_t stride_s_q, const void* mu_p,
                     int64_t stride_mu_p, const void* s_p, int64_t stride_s_p, void* kl, int64_t B, int64_t L,
                     int dtype, vae_stream_t stream);
/* Gradients of the above into contiguous [B, L] buffers; g_mu_p/g_s_p may be NULL. */
int vae_gauss_kl_bwd(const void* mu_q, int64_t stride_mu_q, const void* s_q, int64_t stride_s_q, const void* mu_p,
                     int64_t stride_mu_p, const void* s_p, int64_t stride_s_p, const void* gkl, void* g_mu_q,
                     void* g_s_q, void* g_mu_p, void* g_s_p, int64_t B, int64_t L, int dtype, vae_stream_t stream);

/* ---- VRNN per-step pair ---------------------------------------------------------
 * vrnn.py:419-447 (reparameterize: posterior draw, _clamp_variance :395-408 on the prior
 * logits, prior draw) + vrnn.py:723-733 (kld: KL(post || prior) [+ KL(prior || N(0,1))]),
 * one launch instead of ~5 per time step.
 *   enc_logits, prior_logits [B, 2L] (strided); noise_* [B, L] or NULL (Philox; the prior
 *   uses stream id offset ^ 2^63); z_* [B, L] (NULL allowed when !training);
 *   logvar_post = enc[:, L:] + eps; clamped_prior_logits [B, 2L] = [mu_p | sigmoid(raw)];
 *   logvar_prior = sigmoid(raw) + eps (either logvar output may be NULL: not written); kl_pair [B];
 *   kl_prior [B] or NULL;
 *   means: 4 floats (mu_q, logvar_q, mu_p, logvar_p sums * mean_scale) or NULL.        */
int vae_gauss_pair_fwd(const void* enc_logits, int64_t stride_enc, const void* prior_logits, int64_t stride_prior,
                       const void* noise_post, const void* noise_prior, void* z_post, void* z_prior,
                       void* logvar_post, void* logvar_prior, void* clamped_prior_logits, void* kl_pair,
                       void* kl_prior, float* means, float* workspace, int64_t B, int64_t L, float eps_tol,
                       float mean_scale, int training, uint64_t seed, uint64_t offset, int64_t row0, int dtype,
                       vae_stream_t stream);
/* Grads of every output (each may be NULL) -> g_enc_logits, g_prior_logits [B, 2L] (raw prior logits). */
int vae_gauss_pair_bwd(const void* enc_logits, int64_t stride_enc, const void* prior_logits, int64_t stride_prior,
                       const void* noise_post, const void* noise_prior, const void* gz_post, const void* gz_prior,
                       const void* glogvar_post, const void* glogvar_prior, const void* gclamped, const void* gkl_pair,
                       const void* gkl_prior, const float* gmeans, void* g_enc_logits, void* g_prior_logits,
                       int64_t B, int64_t L, float eps_tol, float mean_scale, int training, uint64_t seed,
                       uint64_t offset, int64_t row0, int dtype, vae_stream_t stream);

/* ---- GumbelSoftmax ----------------------------------------------------------
 * reparameterizers/gumbel.py:110-124 (log_q_z = log_softmax(logits)), :257-278
 * (g = -log(-log(U + 1e-9) + 1e-9); y = softmax((logits + g)/tau + 1e-9); the
 * reference draws U on the CPU and copies it to the device every step), :281-304
 * (h = (y == rowmax(y)), multi-hot on ties; z_hard = (h - y).detach() + y).
 *   logits [B, K] rows `row_stride` apart; unif [B, K] uniform draws or NULL
 *   (Philox keyed by (seed, offset, row0 + row, col)); y, z_hard, log_q_z [B, K], each may be
 *   NULL (not written; at least one is given; log_q_z alone skips the draw and the softmax).
 *   Any K <= VAE_MAX_ROW (K > VAE_MAX_SOFTMAX_ROW: the streamed three-pass kernels).   */
int vae_gumbel_fwd(const void* logits, int64_t row_stride, const void* unif, void* y, int64_t y_stride, void* z_hard,
                   int64_t z_hard_stride, void* log_q_z,
                   int64_t B, int64_t K, float tau, uint64_t seed, uint64_t offset, int64_t row0, int dtype,
                   vae_stream_t stream);
/* glogits [B, K] from the grads of y, z_hard (straight-through: passes to y) and
 * log_q_z (each may be NULL; all NULL writes zeros), using the saved outputs y (needed for gy / gz_hard) and
 * log_q_z (needed for glog_q_z).                                                  */
int vae_gumbel_bwd(const void* y, int64_t y_stride, const void* log_q_z, const void* gy, int64_t gy_stride,
                   const void* gz_hard, int64_t gz_hard_stride, const void* glog_q_z, void* glogits,
                   int64_t glogits_stride, int64_t B, int64_t K, float tau, int dtype, vae_stream_t stream);
/* Categorical KL, reparameterizers/gumbel.py:222-254 (_kl_tf_version over two
 * OneHotCategorical(logits=log_q_z)): kl[b] = sum softmax(q)*(log_softmax(q) -
 * log_softmax(p)).  log_q / log_p: strided [B, K] views; g_log_p may be NULL.     */
int vae_categorical_kl_fwd(const void* log_q, int64_t stride_q, const void* log_p, int64_t stride_p, void* kl,
                           int64_t B, int64_t K, int dtype, vae_stream_t stream);
int vae_categorical_kl_bwd(const void* log_q, int64_t stride_q, const void* log_p, int64_t stride_p, const void* gkl,
                           void* g_log_q, void* g_log_p, int64_t B, int64_t K, int dtype, vae_stream_t stream);

/* ---- ConcatReparameterizer(['isotropic_gaussian', 'discrete']) as one row kernel ------------------------
 * reparameterizers/concat_reparameterizer.py:110-118 (per child: narrow -> forward -> clone, then torch.cat) with
 * the two children of BASELINE config 3, isotropic_gaussian.py:77-131,158-170,204-215 and gumbel.py:110-124,257-304;
 * KL of the pair = the Gaussian child's (the Gumbel child's is identically 0 without a prior, gumbel.py:239-254).
 * A lane group owns the whole contiguous logits row [mu(L) | s(L) | categorical logits(K)] and writes the whole
 * sample row: one launch per direction instead of one per child over strided half rows.
 *   logits [B, 2L+K] rows `row_stride` apart; noise [B, L] / unif [B, K] or NULL (Philox, one (seed, offset) per
 *   child as the per-child entry points take them);
 *   training != 0: z_out [B, L+K] = [z | y]; y, log_q_z may be NULL;
 *   training == 0: z_out = [mu | z_hard]; y [B, K] and log_q_z [B, K] each written when non-NULL;
 *   kl [B].  noise and unif are given together or not at all.  The Gaussian child's two mean diagnostics
 *   (isotropic_gaussian.py:213-214) are not produced here (a grid-wide sum costs this launch 8-12 us): the module
 *   class forms them on first read with torch.mean, as the reference does.
 * Needs L and K multiples of 4 (8 for the 16-bit types), 16-byte aligned bases and strides, and rows that stay
 * register-resident (L/4, K/4 <= 256 vectors): vae_concat_gauss_gumbel_supported tells; VAE_ERR_ARG otherwise
 * (run the children one by one through vae_gauss_reparam_kl_* / vae_gumbel_*, which take the same slices).   */
int vae_concat_gauss_gumbel_supported(int64_t L, int64_t K, int dtype);
int vae_concat_gauss_gumbel_fwd(const void* logits, int64_t row_stride, const void* noise, const void* unif, void* z_out,
                                int64_t z_stride, void* y, void* log_q_z, void* kl, int64_t B, int64_t L, int64_t K,
                                float eps_tol, float tau, int training, uint64_t seed_gauss, uint64_t offset_gauss,
                                uint64_t seed_gumbel, uint64_t offset_gumbel, int64_t row0, int dtype,
                                vae_stream_t stream);
/* Backward of the training-mode forward: glogits [B, 2L+K] from gz [B, L+K] (gradient of the sample row; NULL = none)
 * and gkl [B] (may be NULL); z_out is the forward's sample row (its second part is y).                        */
int vae_concat_gauss_gumbel_bwd(const void* logits, int64_t row_stride, const void* noise, const void* z_out,
                                int64_t z_stride, const void* gz, int64_t gz_stride, const void* gkl, void* glogits,
                                int64_t glogits_stride, int64_t B, int64_t L, int64_t K, float eps_tol, float tau,
                                uint64_t seed_gauss, uint64_t offset_gauss, int64_t row0, int dtype,
                                vae_stream_t stream);

/* ---- Bernoulli (relaxed) ----------------------------------------------------
 * reparameterizers/bernoulli.py:126-136 via torch/distributions/relaxed_bernoulli.py:104-112
 * (p = clamp(sigmoid(l)), u = clamp(U), y = sigmoid((log u - log1p(-u) + log p - log1p(-p))/tau)),
 * bernoulli.py:111-124 (hard = y >= 0.5; z_hard = (hard - y).detach() + y).
 *   logits [B, K] rows `row_stride` apart; unif [B, K] or NULL (Philox); y, z_hard [B, K]
 *   (z_hard may be NULL: not written).                                                    */
int vae_bernoulli_fwd(const void* logits, int64_t row_stride, const void* unif, void* y, int64_t y_stride, void* z_hard,
                      int64_t z_hard_stride, int64_t B,
                      int64_t K, float tau, uint64_t seed, uint64_t offset, int64_t row0, int dtype,
                      vae_stream_t stream);
int vae_bernoulli_bwd(const void* logits, int64_t row_stride, const void* y, int64_t y_stride, const void* gy,
                      int64_t gy_stride, const void* gz_hard, int64_t gz_hard_stride, void* glogits,
                      int64_t glogits_stride, int64_t B, int64_t K, float tau, int dtype, vae_stream_t stream);
/* bernoulli.py:147-191: sum over the last dim of q1*(log(q1+1e-6) - log(p1+1e-6)) +
 * q0*(log(q0+1e-6) - log(p0+1e-6)); logits_p NULL = the uniform prior p = 0.5.     */
int vae_bernoulli_kl_fwd(const void* logits_q, int64_t stride_q, const void* logits_p, int64_t stride_p, void* kl,
                         int64_t B, int64_t K, int dtype, vae_stream_t stream);
int vae_bernoulli_kl_bwd(const void* logits_q, int64_t stride_q, const void* logits_p, int64_t stride_p,
                         const void* gkl, void* g_logits_q, void* g_logits_p, int64_t B, int64_t K, int dtype,
                         vae_stream_t stream);

/* ---- Beta ---------------------------------------------------------------------
 * reparameterizers/beta.py:105-106 (conc1/conc2 = sigmoid of the two halves),
 * :110-124 (KL vs Beta(1.1, 1.1), torch/distributions/kl.py:219-228), :175-176 (means).
 * The sample stays torch (Beta.rsample is an ATen rejection sampler, not injectable).
 *   logits [B, 2L]; conc1, conc2 [B, L]; kl [B]; means: 2 floats * mean_scale or NULL.   */
int vae_beta_split_kl_fwd(const void* logits, int64_t row_stride, void* conc1, void* conc2, void* kl, float* means,
                          float* workspace, int64_t B, int64_t L, float mean_scale, int dtype, vae_stream_t stream);
int vae_beta_split_kl_bwd(const void* logits, int64_t row_stride, const void* gconc1, const void* gconc2,
                          const void* gkl, const float* gmeans, void* glogits, int64_t B, int64_t L, float mean_scale,
                          int dtype, vae_stream_t stream);
/* beta.py:126-136: KL(Beta(a1, a0) || Beta(b1, b0)) summed over the last dim; b1 = b0 = NULL
 * means Beta(1.1, 1.1).  Strided [B, L] views; gb1/gb0 may be NULL.                  */
int vae_beta_kl_fwd(const void* a1, int64_t stride_a1, const void* a0, int64_t stride_a0, const void* b1,
                    int64_t stride_b1, const void* b0, int64_t stride_b0, void* kl, int64_t B, int64_t L, int dtype,
                    vae_stream_t stream);
int vae_beta_kl_bwd(const void* a1, int64_t stride_a1, const void* a0, int64_t stride_a0, const void* b1,
                    int64_t stride_b1, const void* b0, int64_t stride_b0, const void* gkl, void* ga1, void* ga0,
                    void* gb1, void* gb0, int64_t B, int64_t L, int dtype, vae_stream_t stream);

/* ---- Flow "KL" ----------------------------------------------------------------
 * reparameterizers/flow.py:67-78.  cross_batch = 1 (2-D logits, logdet [B,1], R = 1):
 * kl_i = mean_j(sum_f 0.5 u_jf^2 + F/2 log 2pi) - logdet_i -- the reference's [B,B]
 * broadcast in O(B*F).  phase 0 = both steps; phase 1 = row values into kl and
 * their sum * inv_count into mean_io[0] (all-reduce it when the batch is sharded);
 * phase 2 = kl_i = mean_io[0] - logdet_i.  cross_batch = 0 (3-D logits [B,R,F], logdet
 * [B,R]): kl_i = mean_r(nlp_ir - logdet_ir).                                        */
int vae_flow_kl_fwd(const void* u, const void* logdet, int64_t R, void* kl, float* mean_io, float* workspace,
                    int64_t B, int64_t F, float inv_count, int cross_batch, int phase, int dtype,
                    vae_stream_t stream);
/* gu[row, :] = c * u[row, :], c = gsum[0] * scale (cross-batch: gsum = sum_i gkl_i, scale =
 * inv_count) or gkl[row] * scale (per-sample rows of row_len = R*F, scale = 1/R).      */
int vae_flow_kl_bwd(const void* u, const void* gkl, const float* gsum, void* gu, int64_t B, int64_t row_len,
                    float scale, int dtype, vae_stream_t stream);

/* ---- NLL + ELBO -------------------------------------------------------------
 * abstract_vae.py:461-471 (nll -> helpers.distributions.nll, un-vendored: the
 * three definitions are this build's contract, see oracle/cpu_ref.py),
 * abstract_vae.py:356-388 (elbo = nll + kld; loss = (nll + add + beta*kld) - mi;
 * six means), abstract_vae.py:15-37 (variance NLLs read recon rows [mean(D)|var(D)]).
 *   x [B, D]; recon [B, D] (Bernoulli) or [B, 2D]; kld/add/mi [B] or NULL (= 0);
 *   nll [B] always written; loss/elbo [B] both or neither; means: 6 floats
 *   (loss, elbo, nll, kld, add, mi) = sums * inv_count, or NULL.                  */
int vae_nll_elbo_fwd(const void* x, const void* recon, int nll_type, const void* kld, const void* add, const void* mi,
                     float kl_beta, float inv_count, void* nll, void* loss, void* elbo, float* means,
                     float* workspace, int64_t B, int64_t D, int dtype, vae_stream_t stream);
/* Backward: upstream grads of nll/loss/elbo [B] and of the six means (each may be
 * NULL) -> grecon (recon's shape) and the [B] grads of kld/add/mi (each may be NULL). */
int vae_nll_elbo_bwd(const void* x, const void* recon, int nll_type, const void* gnll, const void* gloss,
                     const void* gelbo, const float* gmeans, float kl_beta, float inv_count, void* grecon, void* gkld,
                     void* gadd, void* gmi, int64_t B, int64_t D, int dtype, vae_stream_t stream);
/* Forward AND gradient in one pass over x and recon (no counterpart in the reference, whose autograd
 * re-reads both in its backward): everything vae_nll_elbo_fwd writes, plus
 *   grecon = grad_scale * inv_count * d nll / d recon   (recon's shape),
 * i.e. d(loss_mean)/d(recon) for `loss['loss_mean'].backward()` (abstract_vae.py:381 is what the
 * reference's training loop backpropagates).  loss, elbo, means, workspace are required.
 * vae_nll_fused_grad_finish then multiplies the n stored values by gmeans[0] + gmeans[1] + gmeans[2]
 * (the upstream gradients of loss_mean, elbo_mean, nll_mean; a device array of >= 3 floats) --
 * decided on the device, and a no-op that touches no memory when that sum is exactly 1.
 * The [B]-sized d kld / d add / d mi come from vae_elbo_combine_bwd.                              */
int vae_nll_elbo_fused(const void* x, const void* recon, int nll_type, const void* kld, const void* add, const void* mi,
                       float kl_beta, float inv_count, float grad_scale, void* nll, void* loss, void* elbo, float* means,
                       float* workspace, void* grecon, int64_t B, int64_t D, int dtype, vae_stream_t stream);
int vae_nll_fused_grad_finish(void* grecon, const float* gmeans, int64_t n, int dtype, vae_stream_t stream);
/* vae_nll_elbo_fused for callers that cannot allocate or keep a workspace (code under HIP graph capture, tight
 * serving loops, harnesses that forbid hipMalloc): the same kernels, with the scratch they need -- the closing
 * launch's arrival counters and block partials, 68 KiB (the first 17408 floats of what a caller's workspace holds;
 * the per-chunk row sums that vae_nll_elbo_fused keeps behind them for long rows are not needed: this entry point
 * runs long rows a block per row: for rows of more than 256 * 4 16-byte vectors its nll / loss / elbo equal the
 * workspace form's to rounding -- another summation order -- not bit for bit; shorter rows are bit-identical) -- in a
 * file-scope __device__ array of the library instead of a caller buffer.  No counterpart in the reference.  The call allocates and frees
 * nothing; every GPU the library is used on holds its own copy of the array until the process ends.  Arguments as
 * vae_nll_elbo_fused without `workspace`; the device that owns `stream` must be the current device.  B up to
 * VAE_STATIC_MAX_ROWS, beyond it VAE_ERR_ARG.  All calls on one GPU share that GPU's array: calls that overlap in
 * time on one GPU (two streams, two threads) race on it -- serialise them, or use vae_nll_elbo_fused with a workspace
 * per stream.                                                                                          */
#define VAE_STATIC_MAX_ROWS (1 << 24)
int vae_nll_elbo_fused_static(const void* x, const void* recon, int nll_type, const void* kld, const void* add,
                              const void* mi, float kl_beta, float inv_count, float grad_scale, void* nll, void* loss,
                              void* elbo, float* means, void* grecon, int64_t B, int64_t D, int dtype,
                              vae_stream_t stream);
/* Both of the above in one launch, for the step whose only upstream gradients are those of the six means
 * (loss['loss_mean'].backward(), abstract_vae.py:381): the conditional rescale of the n stored gradient values
 * and gkld / gadd / gmi [B] (each may be NULL) = what vae_elbo_combine_bwd writes with gloss = gelbo = NULL.
 * gmeans: device array of 6 floats.  No counterpart in the reference (its autograd runs ~10 launches here). */
int vae_nll_fused_backward(void* grecon, int64_t n, const float* gmeans, float kl_beta, float inv_count, void* gkld,
                           void* gadd, void* gmi, int64_t B, int dtype, vae_stream_t stream);
/* The ELBO combine alone, from a given nll [B] (K > 1 Monte-Carlo average,
 * abstract_vae.py:347-354; VRNN per-step dicts, vrnn.py:823-829).               */
int vae_elbo_combine_fwd(const void* nll, const void* kld, const void* add, const void* mi, float kl_beta,
                         float inv_count, void* loss, void* elbo, float* means, float* workspace, int64_t B,
                         int dtype, vae_stream_t stream);
int vae_elbo_combine_bwd(const void* gloss, const void* gelbo, const float* gmeans, float kl_beta, float inv_count,
                         void* gnll, void* gkld, void* gadd, void* gmi, int64_t B, int dtype, vae_stream_t stream);

/* ---- multi-sample forms ("same logits, many noises") --------------------------
 * Replace the draw loops of abstract_vae.py:347-354 (loss_function K > 1),
 * msg.py:55-69 (MSGVAE.reparameterize, max_time_steps draws) and
 * abstract_vae.py:301-311 (likelihood): S training-mode draws
 *   z[s] = noise[s] * (0.5*s) + mu,  s = 0..S-1
 * from ONE read of logits; logvar, kl and the two means as in
 * vae_gauss_reparam_kl_fwd (they do not depend on the draw).
 *   noise [S, B, L] or NULL (in-kernel Philox; draw s uses call offset `offset + s`,
 *   i.e. the streams S consecutive single calls would use); z [S, B, L] contiguous. */
int vae_gauss_reparam_multi_fwd(const void* logits, int64_t row_stride, const void* noise, void* z, void* s_out,
                                void* kl, float* means, float* workspace, int64_t B, int64_t L, int64_t S,
                                float eps_tol, float mean_scale, uint64_t seed, uint64_t offset, int64_t row0,
                                int dtype, vae_stream_t stream);
/* gz [S, B, L] or NULL; glogits [B, 2L] contiguous: d mu = sum_s gz[s] + ..., d s = 0.5 sum_s noise[s] gz[s] + ... */
int vae_gauss_reparam_multi_bwd(const void* logits, int64_t row_stride, const void* noise, const void* gz,
                                const void* gs, const void* gkl, const float* gmeans, void* glogits, int64_t B,
                                int64_t L, int64_t S, float eps_tol, float mean_scale, uint64_t seed, uint64_t offset,
                                int64_t row0, int dtype, vae_stream_t stream);
/* S-sample NLL + ELBO (abstract_vae.py:344-354 then :356-388):
 *   nll[b] = (1/S) sum_s nll(x[b], recons[s][b]);  x is read once.
 * recons: HOST array of S device pointers (1 <= S <= VAE_MAX_SAMPLES), each a contiguous
 * [B, D] / [B, 2D] decoder output; everything else as vae_nll_elbo_fwd.            */
int vae_nll_multi_elbo_fwd(const void* x, const void* const* recons, int64_t S, int nll_type, const void* kld,
                           const void* add, const void* mi, float kl_beta, float inv_count, void* nll, void* loss,
                           void* elbo, float* means, float* workspace, int64_t B, int64_t D, int dtype,
                           vae_stream_t stream);
/* grecons: HOST array of S device pointers, each written with d/d recons[s].       */
int vae_nll_multi_elbo_bwd(const void* x, const void* const* recons, void* const* grecons, int64_t S, int nll_type,
                           const void* gnll, const void* gloss, const void* gelbo, const float* gmeans, float kl_beta,
                           float inv_count, void* gkld, void* gadd, void* gmi, int64_t B, int64_t D, int dtype,
                           vae_stream_t stream);

/* ---- VRNN / AdditiveVRNN loss over T time steps ------------------------------------------------
 * vrnn.py:805-829 (loss_function: AbstractVAE.loss_function once per step, the per-step dicts summed by
 * _add_loss_map :767-785 and divided by the step count by _mean_map :787-803), additive_vrnn.py:78-104.
 * The T reconstructions and targets stay where the decoder and the data loader left them: xs / recons are HOST
 * arrays of `steps` device pointers (1 <= steps <= VAE_MAX_STEPS), each a contiguous [B, D] / [B, D | 2D] tensor
 * (a pointer may repeat: one target scored at every step), scored as one [steps * B] batch, step-major:
 *   kld, add, mi [steps*B] (the per-step KL / additional-loss / mutual-information vectors, one after the other:
 *   abstract_vae.py:356-377 runs per step, so VRNN.mut_info :751-765 and the extra terms are per step too), each
 *   may be NULL (= 0);  kl_betas: HOST array of `steps` floats, the KL weight of each step (the reference calls
 *   compute_kl_beta once per step, abstract_vae.py:363: a stateful annealer gives every step its own value), or
 *   NULL = `kl_beta` at every step;
 *   nll, loss, elbo [steps*B]: the per-step vectors;  loss_avg, elbo_avg [B]: their mean over the steps, summed
 *   step by step in the tensor dtype as the reference does (both NULL: not wanted);  means: 6 floats = sums *
 *   inv_count (inv_count = 1 / (steps * B_global): the mean over steps of the per-step means);
 *   elbo_lse [B] or NULL: log-sum-exp over the steps of elbo -- AbstractVAE.likelihood's importance-sampling
 *   estimate (abstract_vae.py:301-315) when the steps are `steps` decoded draws of each sample's posterior.
 * No counterpart launch structure in the reference: ~40 launches per step there.                              */
int vae_nll_steps_elbo_fwd(const void* const* xs, const void* const* recons, int64_t steps, int nll_type,
                           const void* kld, const void* add, const void* mi, const float* kl_betas, float kl_beta,
                           float inv_count, void* nll, void* loss, void* elbo, void* loss_avg, void* elbo_avg,
                           void* elbo_lse, float* means, float* workspace, int64_t B, int64_t D, int dtype,
                           vae_stream_t stream);
/* grecons: HOST array of `steps` device pointers, each written with d / d recons[t] (a reconstruction that is
 * scored at several steps must be given distinct gradient buffers: the caller adds them up);  gnll, gloss,
 * gelbo [steps*B] (each may be NULL), gmeans 6 floats or NULL;  kl_betas / kl_beta as in the forward;
 * gkld, gadd, gmi [steps*B], each may be NULL.                                                              */
int vae_nll_steps_elbo_bwd(const void* const* xs, const void* const* recons, void* const* grecons, int64_t steps,
                           int nll_type, const void* gnll, const void* gloss, const void* gelbo, const float* gmeans,
                           const float* kl_betas, float kl_beta, float inv_count, void* gkld, void* gadd, void* gmi,
                           int64_t B, int64_t D, int dtype, vae_stream_t stream);

/* ---- flow-layer tails (flows/__init__.py; the masked linears / coupling nets stay stock GEMMs) -----
 * Every tail folds FlowSequential's running `logdets += logdet` (flows/__init__.py:555-557):
 * logdet_out[b] = (logdet_in ? logdet_in[b] : 0) + this layer's logdet.  logdet_in may alias
 * logdet_out.  All [B, F] tensors are contiguous unless a stride is given.
 *
 * MADE / MADESplit 'direct' (flows/__init__.py:176-178, :121-125): u = (x - m) * exp(-a),
 * logdet = -sum_f a; pre_exp_tanh applies a = tanh(a) first (:121-122).  m and a are strided views
 * (MADE: the two halves of one [B, 2F] trunk output; MADESplit: two tensors).                      */
int vae_made_tail_fwd(const void* x, int64_t stride_x, const void* m, int64_t stride_m, const void* a, int64_t stride_a,
                      void* u, const void* logdet_in, void* logdet_out, int64_t B, int64_t F, int pre_exp_tanh,
                      int dtype, vae_stream_t stream);
/* The same launch, which also leaves the column moments of u -- what the BatchNormFlow that follows a MADE in every
 * MAF block (flows/__init__.py:582-593) needs first -- as block partials in that layer's `bn_stats` buffer
 * (vae_bnflow_stats_floats(F) floats): vae_bnflow_fwd with training == 2 then skips its own pass over u for the
 * statistics (a MAF block forward moves 24F + 12 bytes per row instead of 28F + 12).  Rows of up to 4 * 256 16-byte
 * vectors, 16-byte aligned (vae_made_tail_moments_supported; VAE_ERR_ARG otherwise: use the two plain calls).   */
int vae_made_tail_moments_supported(int64_t F, int dtype);
int vae_made_tail_fwd_moments(const void* x, int64_t stride_x, const void* m, int64_t stride_m, const void* a,
                              int64_t stride_a, void* u, const void* logdet_in, void* logdet_out, float* bn_stats,
                              int64_t B, int64_t F, int pre_exp_tanh, int dtype, vae_stream_t stream);
/* gu [B, F] / glogdet [B] (each may be NULL) -> gx [B, F] (may be NULL), gm, ga (strided, the raw
 * pre-tanh a when pre_exp_tanh); uses the saved output u.                                        */
int vae_made_tail_bwd(const void* u, const void* a, int64_t stride_a, const void* gu, const void* glogdet, void* gx,
                      void* gm, int64_t stride_gm, void* ga, int64_t stride_ga, int64_t B, int64_t F, int pre_exp_tanh,
                      int dtype, vae_stream_t stream);
/* CouplingLayer 'direct' (flows/__init__.py:524-528): log_s = s_raw*(1-mask), t = t_raw*(1-mask),
 * y = x*exp(log_s) + t, logdet = sum_f log_s.  s_raw NULL = AdditiveCouplingLayer (:458-460:
 * y = x + t, logdet 0).  mask: F floats on the device (16-byte aligned).                         */
int vae_coupling_tail_fwd(const void* x, const void* s_raw, const void* t_raw, const float* mask, void* y,
                          const void* logdet_in, void* logdet_out, int64_t B, int64_t F, int dtype,
                          vae_stream_t stream);
/* gx is the direct d y / d x path only (x also feeds the nets through x*mask: stock autograd). */
int vae_coupling_tail_bwd(const void* x, const void* s_raw, const float* mask, const void* gy, const void* glogdet,
                          void* gx, void* gs_raw, void* gt_raw, int64_t B, int64_t F, int dtype, vae_stream_t stream);
/* BatchNormFlow 'direct' (flows/__init__.py:233-257) [+ the Reverse that follows it in a MAF block,
 * :403-406, when `reverse`: y is stored in reversed column order].
 *   training: batch mean and var = mean((x - mean)^2) + eps over the B rows of THIS call, from one
 *   read of x (shifted sums, block partials added in block order in double); else mean_in / var_in
 *   (the running statistics).  training == 2: as training, but the block partials of x's column moments are
 *   already in `stats`, left there by vae_made_tail_fwd_moments whose output u is this call's x (same B, F).  y = exp(log_gamma) * (x - mean) / sqrt(var) + beta;
 *   logdet = sum_f (log_gamma - 0.5 log var).  log_gamma, beta, mean_in, var_in: F floats.
 *   stats: caller-owned fp32 scratch of vae_bnflow_stats_floats(F) floats, kept for the backward:
 *   [0,F) mean | [F,2F) var | [2F,3F) 1/sqrt(var) | [3F,4F) scale | [4F,5F) beta | [5F,6F) logdet terms |
 *   (backward) [6F,7F) sum_b gy = d beta | [7F,8F) sum_b gy*x_hat | [8F] sum_b glogdet | two [F] vectors of
 *   backward constants | block partials.
 *   The caller updates the running statistics from stats[0,2F) (:240-246) and forms
 *   d log_gamma = exp(log_gamma) * stats[7F..8F) + stats[8F].  F <= VAE_MAX_ROW.                  */
size_t vae_bnflow_stats_floats(int64_t F);
int vae_bnflow_fwd(const void* x, const float* log_gamma, const float* beta, const float* mean_in, const float* var_in,
                   float* stats, void* y, const void* logdet_in, void* logdet_out, int64_t B, int64_t F, float eps,
                   int training, int reverse, int dtype, vae_stream_t stream);
/* gy [B, F] (in y's column order), glogdet [B] or NULL -> gx [B, F]; fills stats[6F, 8F].       */
int vae_bnflow_bwd(const void* x, const void* gy, const void* glogdet, float* stats, void* gx, int64_t B, int64_t F,
                   int training, int reverse, int dtype, vae_stream_t stream);
/* The same backward in two steps, for statistics taken over the rows of SEVERAL ranks (the forward then ran as:
 * vae_bnflow_fwd with y == NULL and training != 0 -- statistics of this rank's rows only, into stats[0,2F) --, the
 * caller's reduction of the moments over ranks, vae_bnflow_fwd with training == 0 on the reduced mean / var):
 *   _sums : stats[6F,7F) = sum_b gy, [7F,8F) = sum_b gy*x_hat, [8F] = sum_b glogdet over this rank's rows;
 *   the caller all-reduces those 2F+1 floats in place (and keeps its local copy for d log_gamma / d beta);
 *   _apply: gx from the reduced sums, `count` = rows over all ranks.
 * No counterpart in the reference (BatchNormFlow there sees one device's batch); SURVEY.md 8e names the
 * cross-rank moments as what a sharded run needs to equal the single-device one.                       */
int vae_bnflow_bwd_sums(const void* x, const void* gy, const void* glogdet, float* stats, int64_t B, int64_t F,
                        int reverse, int dtype, vae_stream_t stream);
int vae_bnflow_bwd_apply(const void* x, const void* gy, float* stats, void* gx, int64_t B, int64_t F, int64_t count,
                         int reverse, int dtype, vae_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VAE_AMD_H_ */
