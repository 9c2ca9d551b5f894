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
 *   - `dtype` selects the tensor element type (VAE_F32 / VAE_BF16); arithmetic and
 *     reductions are always fp32 inside the kernels;
 *   - rows are samples.  `row_stride` / `stride_*` are in ELEMENTS; tensors
 *     without a stride argument are contiguous;
 *   - `workspace` is scratch of at least vae_workspace_bytes() bytes that one call
 *     may use at a time (per stream);
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
enum { VAE_F32 = 0, VAE_BF16 = 1 };
enum { VAE_NLL_BERNOULLI = 0, VAE_NLL_GAUSSIAN = 1, VAE_NLL_DISC_LOGISTIC = 2 };

#define VAE_MAX_ROW (1 << 24) /* longest supported row (elements) */

/* ABI version of this header; bumped on any signature change. */
int vae_abi_version(void);
/* Bytes of scratch every entry point that takes `workspace` may use. */
size_t vae_workspace_bytes(void);

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
int vae_gauss_reparam_kl_fwd(const void* logits, int64_t row_stride, const void* noise, void* z, void* s_out,
                             void* kl, float* means, float* workspace, int64_t B, int64_t L, float eps_tol,
                             float mean_scale, int training, uint64_t seed, uint64_t offset, int64_t row0,
                             int dtype, vae_stream_t stream);
/* Backward of the above: glogits [B, 2L] contiguous from gz [B, L], gs (grad of
 * the logvar output) [B, L], gkl [B], gmeans (2 floats); each grad may be NULL. */
int vae_gauss_reparam_kl_bwd(const void* logits, int64_t row_stride, const void* noise, const void* gz,
                             const void* gs, const void* gkl, const float* gmeans, void* glogits, int64_t B,
                             int64_t L, float eps_tol, float mean_scale, int training, uint64_t seed,
                             uint64_t offset, int64_t row0, int dtype, vae_stream_t stream);

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
/* The ELBO combine alone, from a given nll [B] (K > 1 Monte-Carlo average,
 * abstract_vae.py:347-354; VRNN per-step dicts, vrnn.py:823-829).               */
int vae_elbo_combine_fwd(const void* nll, const void* kld, const void* add, const void* mi, float kl_beta,
                         float inv_count, void* loss, void* elbo, float* means, float* workspace, int64_t B,
                         int dtype, vae_stream_t stream);
int vae_elbo_combine_bwd(const void* gloss, const void* gelbo, const float* gmeans, float kl_beta, float inv_count,
                         void* gnll, void* gkld, void* gadd, void* gmi, int64_t B, int dtype, vae_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VAE_AMD_H_ */
