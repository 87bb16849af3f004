// fs_build.hip -- one-time index-build kernels (not on the per-corpus path).
//
//   k_rownorms   q[v]  = seqsum_d E[v][d]^2          canonical float64, no FMA
//   k_selfdist   d(w)  = 1 - SS / (sqrt(SS)*sqrt(SS)), SS = seq_k q[s_{w+k}]
//                = the CosineDistance (NearPy, reached from
//                /root/reference/search.py:178) of script window w to a fan
//                window with the same vector ids, in the canonical arithmetic
//   k_cmax       max cosine between a script vector and any other table vector:
//                the constant of the exact-n-gram soundness proof
#include "fs_internal.h"

namespace {

__global__ void k_rownorms(const float* __restrict__ emb, uint32_t n_vec, int D,
                           double* __restrict__ q) {
  const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n_vec) return;
  const float* e = emb + (size_t)v * D;
  double acc = 0.0;
  for (int d = 0; d < D; ++d) {
    const double x = (double)e[d];
    acc = __dadd_rn(acc, __dmul_rn(x, x));
  }
  q[v] = acc;
}

// squared norm of an out-of-vocabulary 3-hot vector: number of distinct hot
// positions (sequential sum of 1.0*1.0 terms), positions sorted a <= b <= c
__device__ __forceinline__ double oov_q(uint32_t id, int D) {
  const uint32_t code = id & ~FS_OOV_FLAG;
  const uint32_t c = code % D, b = (code / D) % D, a = code / ((uint32_t)D * D);
  return 1.0 + (b != a ? 1.0 : 0.0) + (c != b ? 1.0 : 0.0);
}

__global__ void k_selfdist(const uint32_t* __restrict__ stok, uint32_t n_windows, int n, int D,
                           const double* __restrict__ q, double* __restrict__ selfdist) {
  const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= n_windows) return;
  double ss = 0.0;
  for (int k = 0; k < n; ++k) {
    const uint32_t id = stok[w + k];
    ss = __dadd_rn(ss, (id & FS_OOV_FLAG) ? oov_q(id, D) : q[id]);
  }
  const double r = __dsqrt_rn(ss);
  selfdist[w] = __dsub_rn(1.0, __ddiv_rn(ss, __dmul_rn(r, r)));
}

__global__ void k_transpose(const float* __restrict__ emb, uint32_t n_vec, int D,
                            const double* __restrict__ q, float* __restrict__ embT) {
  // embT[d][v] = E[v][d] / |E[v]|
  const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n_vec) return;
  const double qq = q[v];
  const float rn = qq > 0.0 ? (float)(1.0 / sqrt(qq)) : 0.0f;
  for (int d = 0; d < D; ++d) embT[(size_t)d * n_vec + v] = emb[(size_t)v * D + d] * rn;
}

constexpr int kCmaxU = 32;   // script rows per block, staged in LDS

__global__ __launch_bounds__(256) void k_cmax(const float* __restrict__ emb,
                                              const float* __restrict__ embT, uint32_t n_vec,
                                              int D, const double* __restrict__ q,
                                              const uint32_t* __restrict__ rows_u, uint32_t n_u,
                                              int* __restrict__ out_bits) {
  extern __shared__ float s_u[];   // [kCmaxU][D], unit-normalised
  const uint32_t u0 = blockIdx.y * kCmaxU;
  for (uint32_t i = threadIdx.x; i < (uint32_t)kCmaxU * D; i += blockDim.x) {
    const uint32_t ui = u0 + i / D;
    float val = 0.0f;
    if (ui < n_u) {
      const uint32_t row = rows_u[ui];
      const double qq = q[row];
      const float rn = qq > 0.0 ? (float)(1.0 / sqrt(qq)) : 0.0f;
      val = emb[(size_t)row * D + i % D] * rn;
    }
    s_u[i] = val;
  }
  __syncthreads();
  const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
  float best = 0.0f;
  if (v < n_vec) {
    float acc[kCmaxU];
#pragma unroll
    for (int i = 0; i < kCmaxU; ++i) acc[i] = 0.0f;
    for (int d = 0; d < D; ++d) {
      const float x = embT[(size_t)d * n_vec + v];
#pragma unroll
      for (int i = 0; i < kCmaxU; ++i) acc[i] = fmaf(s_u[i * D + d], x, acc[i]);
    }
#pragma unroll
    for (int i = 0; i < kCmaxU; ++i) {
      const uint32_t ui = u0 + i;
      if (ui < n_u && rows_u[ui] != v && acc[i] > best) best = acc[i];
    }
  }
  for (int d = 32; d > 0; d >>= 1) {
    const float o = __shfl_xor(best, d);
    best = o > best ? o : best;
  }
  if ((threadIdx.x & 63) == 0 && best > 0.0f) atomicMax(out_bits, __float_as_int(best));
}

// Pairs (script vector u, table vector v) that are *near*: cos(u, v) > 1 - coef / (|u| |v|), with
// coef = n * thr * a_max^2 / 2 (fs_lsh.hip, component ids): same tiling as k_cmax, the pairs
// appended to a list (float32 cosines, 1e-4 of slack towards more pairs).  With gamma > -1.5 the
// relation is angular instead: cos(u, v) > gamma, and a vector of norm 0 is near nothing (the
// share rule of fs_lsh.hip, where a far slot's dot product is bounded by gamma |u| |v|: 0 for it).
__global__ __launch_bounds__(256) void k_near_pairs(const float* __restrict__ emb,
                                                    const float* __restrict__ embT, uint32_t n_vec,
                                                    int D, const double* __restrict__ q,
                                                    const uint32_t* __restrict__ rows_u, uint32_t n_u,
                                                    float coef, float gamma, uint2* __restrict__ pairs, uint32_t cap,
                                                    uint32_t* __restrict__ count) {
  extern __shared__ float s_u[];   // [kCmaxU][D], unit-normalised
  __shared__ float s_nu[kCmaxU];
  const uint32_t u0 = blockIdx.y * kCmaxU;
  for (uint32_t i = threadIdx.x; i < (uint32_t)kCmaxU * D; i += blockDim.x) {
    const uint32_t ui = u0 + i / D;
    float val = 0.0f;
    if (ui < n_u) {
      const uint32_t row = rows_u[ui];
      const double qq = q[row];
      const float rn = qq > 0.0 ? (float)(1.0 / sqrt(qq)) : 0.0f;
      val = emb[(size_t)row * D + i % D] * rn;
    }
    s_u[i] = val;
  }
  if (threadIdx.x < kCmaxU) s_nu[threadIdx.x] = u0 + threadIdx.x < n_u ? (float)sqrt(q[rows_u[u0 + threadIdx.x]]) : 0.0f;
  __syncthreads();
  const uint32_t v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v >= n_vec) return;
  float acc[kCmaxU];
#pragma unroll
  for (int i = 0; i < kCmaxU; ++i) acc[i] = 0.0f;
  for (int d = 0; d < D; ++d) {
    const float x = embT[(size_t)d * n_vec + v];
#pragma unroll
    for (int i = 0; i < kCmaxU; ++i) acc[i] = fmaf(s_u[i * D + d], x, acc[i]);
  }
  const float nv = (float)sqrt(q[v]);
#pragma unroll
  for (int i = 0; i < kCmaxU; ++i) {
    const uint32_t ui = u0 + i;
    if (ui >= n_u || rows_u[ui] == v) continue;
    const float nn = s_nu[i] * nv;
    // (a vector of norm 0 is near everything: its slot adds nothing to a window's cosine)
    const bool near = gamma > -1.5f ? (nn > 0.0f && acc[i] > gamma - 1e-4f)
                                    : (!(nn > 0.0f) || acc[i] > 1.0f - coef / nn - 1e-4f);
    if (near) {
      const uint32_t at = atomicAdd(count, 1u);
      if (at < cap) pairs[at] = make_uint2(rows_u[ui], v);
    }
  }
}

// max over the script's vectors u (rows_u; nullptr: the n_u rows of the table) of max_d |u[d]| / |u| (the share rule of fs_lsh.hip: the cosine of a
// vector with at most three non-zero coordinates, all 1, to u is at most sqrt(3) times that)
__global__ void k_coordmax(const float* __restrict__ emb, int D, const double* __restrict__ q,
                           const uint32_t* __restrict__ rows_u, uint32_t n_u, int* __restrict__ out_bits) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_u) return;
  const uint32_t row = rows_u ? rows_u[i] : i;     // (no list: every row of the table)
  const double qq = q[row];
  if (!(qq > 0.0)) return;
  float m = 0.0f;
  for (int d = 0; d < D; ++d) m = fmaxf(m, fabsf(emb[(size_t)row * D + d]));
  const float r = (float)((double)m / sqrt(qq)) * (1.0f + 1e-6f);
  atomicMax(out_bits, __float_as_int(r));
}

}  // namespace

int fs_launch_coordmax(const float* emb, int D, const uint32_t* rows_u, uint32_t n_u, const double* q,
                       int* d_out_bits, hipStream_t s) {
  if (!n_u) return FS_OK;
  hipLaunchKernelGGL(k_coordmax, dim3((n_u + 255) / 256), dim3(256), 0, s, emb, D, q, rows_u, n_u, d_out_bits);
  FS_HIP(hipGetLastError());
  return FS_OK;
}

int fs_launch_near_pairs(const float* emb, uint64_t n_vec, int D, const uint32_t* rows_u, uint32_t n_u,
                         const double* q, float* embT_scratch, float coef, float gamma, uint2* pairs, uint32_t cap,
                         uint32_t* count, hipStream_t s) {
  if (!n_vec || !n_u) return FS_OK;
  hipLaunchKernelGGL(k_transpose, dim3((uint32_t)((n_vec + 255) / 256)), dim3(256), 0, s, emb,
                     (uint32_t)n_vec, D, q, embT_scratch);
  const size_t lds = (size_t)kCmaxU * D * sizeof(float);
  FS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_near_pairs),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  dim3 grid((uint32_t)((n_vec + 255) / 256), (n_u + kCmaxU - 1) / kCmaxU);
  hipLaunchKernelGGL(k_near_pairs, grid, dim3(256), lds, s, emb, embT_scratch, (uint32_t)n_vec, D, q,
                     rows_u, n_u, coef, gamma, pairs, cap, count);
  FS_HIP(hipGetLastError());
  return FS_OK;
}

int fs_launch_rownorms(const float* emb, uint64_t n_vec, int D, double* q, hipStream_t s) {
  if (!n_vec) return FS_OK;
  hipLaunchKernelGGL(k_rownorms, dim3((uint32_t)((n_vec + 255) / 256)), dim3(256), 0, s, emb,
                     (uint32_t)n_vec, D, q);
  FS_HIP(hipGetLastError());
  return FS_OK;
}

int fs_launch_selfdist(const uint32_t* stok, uint64_t n_windows, int n, int D, uint64_t n_vec,
                       const double* q, double* selfdist, hipStream_t s) {
  (void)n_vec;
  if (!n_windows) return FS_OK;
  hipLaunchKernelGGL(k_selfdist, dim3((uint32_t)((n_windows + 255) / 256)), dim3(256), 0, s, stok,
                     (uint32_t)n_windows, n, D, q, selfdist);
  FS_HIP(hipGetLastError());
  return FS_OK;
}

int fs_launch_cmax(const float* emb, uint64_t n_vec, int D, const uint32_t* rows_u, uint32_t n_u,
                   const double* q, float* embT_scratch, int* d_out_bits, hipStream_t s) {
  if (!n_vec || !n_u) return FS_OK;
  hipLaunchKernelGGL(k_transpose, dim3((uint32_t)((n_vec + 255) / 256)), dim3(256), 0, s, emb,
                     (uint32_t)n_vec, D, q, embT_scratch);
  const size_t lds = (size_t)kCmaxU * D * sizeof(float);
  FS_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_cmax),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  dim3 grid((uint32_t)((n_vec + 255) / 256), (n_u + kCmaxU - 1) / kCmaxU);
  hipLaunchKernelGGL(k_cmax, grid, dim3(256), lds, s, emb, embT_scratch, (uint32_t)n_vec, D, q,
                     rows_u, n_u, d_out_bits);
  FS_HIP(hipGetLastError());
  return FS_OK;
}
