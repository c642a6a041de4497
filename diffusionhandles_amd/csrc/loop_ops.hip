// Small f32 kernels of the denoising loops: CFG combine + DDIM step, latent update,
// Adam step on the null-text embedding, MSE and its gradient.
#include "common.h"

namespace dh {

__global__ void k_ddim_cfg(float* out, const float* x, const float* eu, const float* ec, float scale, float sa_t,
                           float s1a_t, float sa_p, float s1a_p, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float e = ec[i];
  if (eu) { float u = eu[i]; e = u + scale * (e - u); }
  float x0 = (x[i] - s1a_t * e) / sa_t;
  out[i] = sa_p * x0 + s1a_p * e;
}

__global__ void k_latent_update(float* out, const float* x, const float* g, float k, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = x[i] - k * g[i];
}

__global__ void k_adam(float* p, const float* g, float* m, float* v, float lr, float b1, float b2, float eps,
                       float bc1, float bc2_sqrt, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float gi = g[i];
  float mi = m[i] + (1.f - b1) * (gi - m[i]);          // lerp form used by torch
  float vi = b2 * v[i] + (1.f - b2) * gi * gi;
  m[i] = mi;
  v[i] = vi;
  float denom = sqrtf(vi) / bc2_sqrt + eps;
  p[i] = p[i] - (lr / bc1) * (mi / denom);
}

// one workgroup: the latent has 16 K - 300 K elements, and a single-block reduction needs no partials buffer
// (nothing process-global, legal inside stream capture, deterministic summation order)
__global__ void __launch_bounds__(1024) k_mse(const float* a, const float* b, int n, float* out, float* d_a) {
  __shared__ double sm[16];
  double l = 0.0;
  const float k = 2.f / (float)n;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    float d = a[i] - b[i];
    l += (double)d * (double)d;
    if (d_a) d_a[i] = k * d;
  }
  l = block_sum(l, sm);
  if (threadIdx.x == 0) out[0] = (float)(l / (double)n);
}

}  // namespace dh
using namespace dh;

extern "C" int dh_ddim_cfg_step(float* x_out, const float* x, const float* eps_u, const float* eps_c, float scale,
                                float alpha_t, float alpha_prev, int n, void* stream) {
  DH_REQUIRE(x_out && x && eps_c && n > 0, "bad arguments");
  // torch evaluates a**0.5 on float32 0-d tensors: float32 sqrt of the float32 alpha
  float sa_t = sqrtf(alpha_t), s1a_t = sqrtf(1.f - alpha_t), sa_p = sqrtf(alpha_prev), s1a_p = sqrtf(1.f - alpha_prev);
  hipLaunchKernelGGL(k_ddim_cfg, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x_out, x, eps_u, eps_c, scale,
                     sa_t, s1a_t, sa_p, s1a_p, n);
  DH_LAUNCH_CHECK();
  return DH_OK;
}

extern "C" int dh_latent_update(float* x_out, const float* x, const float* g, float lr, float grad_scale, int n,
                                void* stream) {
  DH_REQUIRE(x_out && x && g && n > 0 && grad_scale != 0.f, "bad arguments");
  hipLaunchKernelGGL(k_latent_update, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, x_out, x, g,
                     lr / grad_scale, n);
  DH_LAUNCH_CHECK();
  return DH_OK;
}

extern "C" int dh_adam_step(float* p, const float* g, float* m, float* v, float lr, float beta1, float beta2,
                            float eps, int step, int n, void* stream) {
  DH_REQUIRE(p && g && m && v && n > 0 && step >= 1, "bad arguments");
  float bc1 = 1.f - powf(beta1, (float)step);
  float bc2 = sqrtf(1.f - powf(beta2, (float)step));
  hipLaunchKernelGGL(k_adam, dim3(cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, lr, beta1, beta2, eps,
                     bc1, bc2, n);
  DH_LAUNCH_CHECK();
  return DH_OK;
}

extern "C" int dh_mse_fwd_bwd(const float* rec, const float* target, int n, float* loss_out, float* d_rec,
                              void* stream) {
  DH_REQUIRE(rec && target && loss_out && n > 0, "bad arguments");
  hipLaunchKernelGGL(k_mse, dim3(1), dim3(1024), 0, (hipStream_t)stream, rec, target, n, loss_out, d_rec);
  DH_LAUNCH_CHECK();
  return DH_OK;
}
