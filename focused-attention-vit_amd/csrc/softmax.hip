// Row softmax (+ mask, + dropout) for the dense attention variants: vit.MultiHeadAttention
// (models/vit.py:96-97), CrossAttention / MultiHeadCrossAttention (models/attention.py:67-72,
// 134-141) and the nn.MultiheadAttention fallback.  The QK^T and attn.V contractions run on
// the MFMA GEMM (gemm.hip, batched); this kernel is the HBM-bound piece in between:
// one 64-lane wave per score row, shuffle reductions for row max / row sum.
#include "common.h"

namespace {

template <typename PT>
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ S, PT* __restrict__ P,
                                                          PT* __restrict__ Pd, const uint8_t* __restrict__ mask,
                                                          long m_sb, long m_sq, int H, long rows, int Lq, int Lk,
                                                          uint32_t thresh, float keep_scale, uint64_t seed,
                                                          const unsigned long long* epoch) {
  if (thresh) seed = favit_eff_seed(seed, epoch);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long row = (long)blockIdx.x * 4 + wave;
  if (row >= rows) return;
  const long z = row / Lq;
  const int q = (int)(row - z * Lq);
  const uint8_t* mrow = mask ? mask + (z / H) * m_sb + (long)q * m_sq : nullptr;
  const float* s = S + row * (long)Lk;
  float m = -INFINITY;
  for (int k = lane; k < Lk; k += 64) {
    const float v = (mrow && mrow[k] == 0) ? -INFINITY : s[k];
    m = fmaxf(m, v);
  }
  m = wave_max(m);
  float l = 0.f;
  for (int k = lane; k < Lk; k += 64) {
    const float v = (mrow && mrow[k] == 0) ? -INFINITY : s[k];
    l += __expf(v - m);
  }
  l = wave_sum(l);
  const float inv = 1.0f / l;
  for (int k = lane; k < Lk; k += 64) {
    const float v = (mrow && mrow[k] == 0) ? -INFINITY : s[k];
    const float p = __expf(v - m) * inv;
    P[row * (long)Lk + k] = from_f32<PT>(p);
    if (Pd) {
      const bool keep = favit_keep(seed, (uint64_t)(row * (long)Lk + k), thresh);
      Pd[row * (long)Lk + k] = from_f32<PT>(keep ? p * keep_scale : 0.f);
    }
  }
}

template <typename PT, typename ST>
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const PT* __restrict__ P, const float* __restrict__ dPd,
                                                          ST* __restrict__ dS, long rows, int Lk, uint32_t thresh,
                                                          float keep_scale, uint64_t seed, const unsigned long long* epoch) {
  if (thresh) seed = favit_eff_seed(seed, epoch);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long row = (long)blockIdx.x * 4 + wave;
  if (row >= rows) return;
  const long off = row * (long)Lk;
  float dot = 0.f;
  for (int k = lane; k < Lk; k += 64) {
    float dp = dPd[off + k];
    if (thresh) dp = favit_keep(seed, (uint64_t)(off + k), thresh) ? dp * keep_scale : 0.f;
    dot = fmaf(to_f32(P[off + k]), dp, dot);
  }
  dot = wave_sum(dot);
  for (int k = lane; k < Lk; k += 64) {
    float dp = dPd[off + k];
    if (thresh) dp = favit_keep(seed, (uint64_t)(off + k), thresh) ? dp * keep_scale : 0.f;
    dS[off + k] = from_f32<ST>(to_f32(P[off + k]) * (dp - dot));
  }
}

}  // namespace

extern "C" int favit_softmax_fwd(const float* S, void* P, void* Pd, int p_dtype, const uint8_t* mask, int64_t m_sb,
                                 int64_t m_sq, int32_t H, int64_t Z, int32_t Lq, int32_t Lk, float dropout_p,
                                 uint64_t seed, void* stream) {
  if (!S || !P || Z <= 0 || Lq <= 0 || Lk <= 0 || H <= 0 || dropout_p < 0.f || dropout_p >= 1.f)
    return FAVIT_ERR_INVALID;
  if (dropout_p > 0.f && !Pd) return FAVIT_ERR_INVALID;
  const long rows = (long)Z * Lq;
  const uint32_t th = dropout_threshold(dropout_p);
  const float ks = 1.0f / (1.0f - dropout_p);
  const dim3 grid((unsigned)((rows + 3) / 4));
  hipStream_t st = as_stream(stream);
  if (p_dtype == FAVIT_F32)
    hipLaunchKernelGGL((softmax_fwd_kernel<float>), grid, dim3(256), 0, st, S, (float*)P, (float*)(th ? Pd : nullptr), mask,
                       (long)m_sb, (long)m_sq, H, rows, Lq, Lk, th, ks, seed, favit_dropout_epoch_ptr_());
  else if (p_dtype == FAVIT_BF16)
    hipLaunchKernelGGL((softmax_fwd_kernel<bf16_t>), grid, dim3(256), 0, st, S, (bf16_t*)P, (bf16_t*)(th ? Pd : nullptr),
                       mask, (long)m_sb, (long)m_sq, H, rows, Lq, Lk, th, ks, seed, favit_dropout_epoch_ptr_());
  else
    return FAVIT_ERR_INVALID;
  FAVIT_CHECK_LAUNCH();
  return FAVIT_OK;
}

extern "C" int favit_softmax_bwd(const void* P, int p_dtype, const float* dPd, void* dS, int ds_dtype, int64_t Z,
                                 int32_t Lq, int32_t Lk, float dropout_p, uint64_t seed, void* stream) {
  if (!P || !dPd || !dS || Z <= 0 || Lq <= 0 || Lk <= 0 || dropout_p < 0.f || dropout_p >= 1.f)
    return FAVIT_ERR_INVALID;
  if (p_dtype != ds_dtype) return FAVIT_ERR_UNSUPPORTED;
  const long rows = (long)Z * Lq;
  const uint32_t th = dropout_threshold(dropout_p);
  const float ks = 1.0f / (1.0f - dropout_p);
  const dim3 grid((unsigned)((rows + 3) / 4));
  hipStream_t st = as_stream(stream);
  if (p_dtype == FAVIT_F32)
    hipLaunchKernelGGL((softmax_bwd_kernel<float, float>), grid, dim3(256), 0, st, (const float*)P, dPd, (float*)dS, rows,
                       Lk, th, ks, seed, favit_dropout_epoch_ptr_());
  else if (p_dtype == FAVIT_BF16)
    hipLaunchKernelGGL((softmax_bwd_kernel<bf16_t, bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)P, dPd, (bf16_t*)dS,
                       rows, Lk, th, ks, seed, favit_dropout_epoch_ptr_());
  else
    return FAVIT_ERR_INVALID;
  FAVIT_CHECK_LAUNCH();
  return FAVIT_OK;
}
