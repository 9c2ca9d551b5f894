/*
 * A torch-free host for the C ABI: plain C, the HIP runtime API for memory, dlopen() for libvae_amd.so.
 * Runs one ELBO step of the hot path (IsotropicGaussian sample + KL -> Bernoulli NLL + ELBO with the gradient in
 * the same pass -> loss backward -> reparameterization backward) and checks every output against a scalar
 * restatement of the reference's arithmetic written out below (isotropic_gaussian.py:77-131,158-170;
 * abstract_vae.py:356-387; Bernoulli-with-logits NLL).  Test infrastructure: built by __graft_entry__.build(),
 * run by tests/test_gpu_abi_client.py.
 *
 *   usage: elbo_step_client /path/to/libvae_amd.so
 */
#include <dlfcn.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "../../include/vae_amd.h"

#define CHECK_HIP(e)                                                               \
  do {                                                                             \
    hipError_t err_ = (e);                                                         \
    if (err_ != hipSuccess) {                                                      \
      fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(err_)); \
      return 2;                                                                    \
    }                                                                              \
  } while (0)

typedef int (*abi_version_fn)(void);
typedef size_t (*ws_bytes_fn)(void);
typedef int (*gauss_fwd_fn)(const void*, int64_t, const void*, void*, int64_t, void*, void*, float*, float*, int64_t,
                            int64_t, float, float, int, uint64_t, uint64_t, int64_t, int, vae_stream_t);
typedef int (*gauss_bwd_fn)(const void*, int64_t, const void*, const void*, int64_t, const void*, const void*,
                            const float*, void*, int64_t, int64_t, int64_t, float, float, int, uint64_t, uint64_t,
                            int64_t, int, vae_stream_t);
typedef int (*nll_fused_fn)(const void*, const void*, int, const void*, const void*, const void*, float, float, float,
                            void*, void*, void*, float*, float*, void*, int64_t, int64_t, int, vae_stream_t);
typedef int (*fused_bwd_fn)(void*, int64_t, const float*, float, float, void*, void*, void*, int64_t, int,
                            vae_stream_t);

static uint32_t lcg_state = 12345u;
static float uniform01(void) {
  lcg_state = lcg_state * 1664525u + 1013904223u;
  return (float)((lcg_state >> 8) + 0.5) / 16777216.0f;
}
static float normal(void) {
  const float u1 = uniform01(), u2 = uniform01();
  return sqrtf(-2.0f * logf(u1)) * cosf(6.2831853f * u2);
}

static int close_enough(const char* what, double got, double want, double rtol, double atol, int idx) {
  if (fabs(got - want) <= atol + rtol * fabs(want)) return 1;
  fprintf(stderr, "MISMATCH %s[%d]: got %.9g want %.9g\n", what, idx, got, want);
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: %s libvae_amd.so\n", argv[0]);
    return 2;
  }
  void* lib = dlopen(argv[1], RTLD_NOW | RTLD_LOCAL);
  if (!lib) {
    fprintf(stderr, "dlopen: %s\n", dlerror());
    return 2;
  }
  abi_version_fn abi_version = (abi_version_fn)dlsym(lib, "vae_abi_version");
  ws_bytes_fn ws_bytes = (ws_bytes_fn)dlsym(lib, "vae_workspace_bytes");
  gauss_fwd_fn gauss_fwd = (gauss_fwd_fn)dlsym(lib, "vae_gauss_reparam_kl_fwd");
  gauss_bwd_fn gauss_bwd = (gauss_bwd_fn)dlsym(lib, "vae_gauss_reparam_kl_bwd");
  nll_fused_fn nll_fused = (nll_fused_fn)dlsym(lib, "vae_nll_elbo_fused");
  fused_bwd_fn fused_bwd = (fused_bwd_fn)dlsym(lib, "vae_nll_fused_backward");
  if (!abi_version || !ws_bytes || !gauss_fwd || !gauss_bwd || !nll_fused || !fused_bwd) {
    fprintf(stderr, "missing symbol: %s\n", dlerror());
    return 2;
  }

  enum { B = 67, L = 36, D = 784 };
  const float eps_tol = 1e-6f, beta = 0.7f;
  const float inv_count = 1.0f / B, mean_scale = 1.0f / (B * L);
  static float logits[B * 2 * L], noise[B * L], x[B * D], recon[B * D], gz[B * L];
  for (int i = 0; i < B * 2 * L; ++i) logits[i] = normal();
  for (int i = 0; i < B * L; ++i) noise[i] = normal();
  for (int i = 0; i < B * L; ++i) gz[i] = normal() * inv_count;
  for (int i = 0; i < B * D; ++i) x[i] = uniform01() < 0.13f ? 1.0f : 0.0f;
  for (int i = 0; i < B * D; ++i) recon[i] = 2.0f * normal();

  float *d_logits, *d_noise, *d_x, *d_recon, *d_gz, *d_z, *d_kl, *d_nll, *d_loss, *d_elbo, *d_means, *d_gmeans,
      *d_grecon, *d_gkld, *d_glogits, *d_ws;
  CHECK_HIP(hipSetDevice(0));
  CHECK_HIP(hipMalloc((void**)&d_logits, sizeof(logits)));
  CHECK_HIP(hipMalloc((void**)&d_noise, sizeof(noise)));
  CHECK_HIP(hipMalloc((void**)&d_x, sizeof(x)));
  CHECK_HIP(hipMalloc((void**)&d_recon, sizeof(recon)));
  CHECK_HIP(hipMalloc((void**)&d_gz, sizeof(gz)));
  CHECK_HIP(hipMalloc((void**)&d_z, sizeof(noise)));
  CHECK_HIP(hipMalloc((void**)&d_kl, B * sizeof(float)));
  CHECK_HIP(hipMalloc((void**)&d_nll, B * sizeof(float)));
  CHECK_HIP(hipMalloc((void**)&d_loss, B * sizeof(float)));
  CHECK_HIP(hipMalloc((void**)&d_elbo, B * sizeof(float)));
  CHECK_HIP(hipMalloc((void**)&d_means, 8 * sizeof(float)));
  CHECK_HIP(hipMalloc((void**)&d_gmeans, 6 * sizeof(float)));
  CHECK_HIP(hipMalloc((void**)&d_grecon, sizeof(recon)));
  CHECK_HIP(hipMalloc((void**)&d_gkld, B * sizeof(float)));
  CHECK_HIP(hipMalloc((void**)&d_glogits, sizeof(logits)));
  CHECK_HIP(hipMalloc((void**)&d_ws, ws_bytes()));
  CHECK_HIP(hipMemset(d_ws, 0, ws_bytes())); /* zeroed once, as the header asks */
  CHECK_HIP(hipMemset(d_means, 0, 8 * sizeof(float)));
  CHECK_HIP(hipMemcpy(d_logits, logits, sizeof(logits), hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(d_noise, noise, sizeof(noise), hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(d_x, x, sizeof(x), hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(d_recon, recon, sizeof(recon), hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(d_gz, gz, sizeof(gz), hipMemcpyHostToDevice));
  const float gmeans[6] = {1.0f, 0, 0, 0, 0, 0}; /* loss_mean.backward() */
  CHECK_HIP(hipMemcpy(d_gmeans, gmeans, sizeof(gmeans), hipMemcpyHostToDevice));

  hipStream_t st;
  CHECK_HIP(hipStreamCreate(&st));
  int rc = 0;
  rc |= gauss_fwd(d_logits, 2 * L, d_noise, d_z, L, NULL, d_kl, d_means + 6, d_ws, B, L, eps_tol, mean_scale, 1, 0, 0,
                  0, VAE_F32, st);
  rc |= nll_fused(d_x, d_recon, VAE_NLL_BERNOULLI, d_kl, NULL, NULL, beta, inv_count, 1.0f, d_nll, d_loss, d_elbo,
                  d_means, d_ws, d_grecon, B, D, VAE_F32, st);
  rc |= fused_bwd(d_grecon, (int64_t)B * D, d_gmeans, beta, inv_count, d_gkld, NULL, NULL, B, VAE_F32, st);
  rc |= gauss_bwd(d_logits, 2 * L, d_noise, d_gz, L, NULL, d_gkld, NULL, d_glogits, 2 * L, B, L, eps_tol, mean_scale,
                  1, 0, 0, 0, VAE_F32, st);
  if (rc != VAE_OK) {
    fprintf(stderr, "an entry point returned %d\n", rc);
    return 1;
  }
  CHECK_HIP(hipStreamSynchronize(st));

  static float z[B * L], kl[B], nll[B], loss[B], elbo[B], means[8], grecon[B * D], gkld[B], glogits[B * 2 * L];
  CHECK_HIP(hipMemcpy(z, d_z, sizeof(z), hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(kl, d_kl, sizeof(kl), hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(nll, d_nll, sizeof(nll), hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(loss, d_loss, sizeof(loss), hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(elbo, d_elbo, sizeof(elbo), hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(means, d_means, sizeof(means), hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(grecon, d_grecon, sizeof(grecon), hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(gkld, d_gkld, sizeof(gkld), hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(glogits, d_glogits, sizeof(glogits), hipMemcpyDeviceToHost));

  /* the reference's arithmetic, one sample at a time */
  int ok = 1;
  double sum_loss = 0, sum_elbo = 0, sum_nll = 0, sum_kl = 0, sum_mu = 0, sum_s = 0;
  for (int b = 0; b < B && ok; ++b) {
    double k = 0.0;
    for (int j = 0; j < L; ++j) {
      const float mu = logits[b * 2 * L + j];
      const float s = logits[b * 2 * L + L + j] + eps_tol;
      /* z = e * (0.5 * s) + mu with every fp32 op rounded on its own, the reference's order: bit-exact */
      volatile float half_s = 0.5f * s;
      volatile float prod = noise[b * L + j] * half_s;
      const float want_z_unfused = prod + mu;
      if (z[b * L + j] != want_z_unfused) {
        fprintf(stderr, "MISMATCH z[%d,%d]: got %.9g want %.9g\n", b, j, z[b * L + j], want_z_unfused);
        ok = 0;
        break;
      }
      k += 0.5 * ((double)s * s + (double)mu * mu - 1.0 - log((double)s * s));
      sum_mu += mu;
      sum_s += s;
      /* d/dmu = gz + gkl*mu ; d/ds = 0.5*e*gz + gkl*(s - 1/s), gkl = beta/B */
      const double gk = (double)beta * inv_count;
      ok &= close_enough("glogits_mu", glogits[b * 2 * L + j], gz[b * L + j] + gk * mu, 1e-5, 1e-8, b * L + j);
      ok &= close_enough("glogits_s", glogits[b * 2 * L + L + j],
                         0.5 * noise[b * L + j] * gz[b * L + j] + gk * ((double)s - 1.0 / (double)s), 1e-5, 1e-7,
                         b * L + j);
    }
    double n = 0.0;
    for (int d = 0; d < D; ++d) {
      const double l = recon[b * D + d], t = x[b * D + d];
      n += fmax(l, 0.0) - l * t + log1p(exp(-fabs(l)));
      ok &= close_enough("grecon", grecon[b * D + d], (1.0 / (1.0 + exp(-l)) - t) * inv_count, 1e-5, 2e-7 * inv_count,
                         b * D + d);
    }
    ok &= close_enough("kl", kl[b], k, 1e-5, 1e-4, b);
    ok &= close_enough("nll", nll[b], n, 1e-5, 1e-3, b);
    ok &= close_enough("elbo", elbo[b], (double)nll[b] + kl[b], 1e-6, 0, b);
    ok &= close_enough("loss", loss[b], (double)nll[b] + (double)beta * kl[b], 1e-6, 0, b);
    ok &= close_enough("gkld", gkld[b], (double)beta * inv_count, 1e-6, 0, b);
    sum_loss += loss[b];
    sum_elbo += elbo[b];
    sum_nll += nll[b];
    sum_kl += kl[b];
  }
  if (ok) {
    ok &= close_enough("loss_mean", means[0], sum_loss / B, 1e-6, 0, 0);
    ok &= close_enough("elbo_mean", means[1], sum_elbo / B, 1e-6, 0, 0);
    ok &= close_enough("nll_mean", means[2], sum_nll / B, 1e-6, 0, 0);
    ok &= close_enough("kld_mean", means[3], sum_kl / B, 1e-6, 0, 0);
    ok &= close_enough("mu_mean", means[6], sum_mu / (B * L), 1e-5, 1e-6, 0);
    ok &= close_enough("logvar_mean", means[7], sum_s / (B * L), 1e-5, 1e-6, 0);
  }
  printf("abi %d, B=%d L=%d D=%d, loss_mean %.6f: %s\n", abi_version(), B, L, D, means[0], ok ? "OK" : "FAILED");
  return ok ? 0 : 1;
}
