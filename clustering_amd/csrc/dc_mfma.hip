// dc_mfma.hip -- host side of the MFMA variants: workspace layout, operand-image kernels and the
// switch over the per-K-step translation units (kernels: dc_mfma_kernels.hpp).
#include "dc_mfma_kernels.hpp"

#include <algorithm>

#ifndef DC_STEP_MASK
#define DC_STEP_MASK 0xFFFFu   // bit (n-1) set <=> dc_mfma_step.hip was built with -DDC_STEP=n
#endif

namespace dc {

namespace {

// ---------------------------------------------------------------------------------------------
// operand images
// ---------------------------------------------------------------------------------------------
__global__ void colsum_kernel(const float* __restrict__ coords, uint32_t n_rows, uint32_t D,
                              double* __restrict__ sums) {
  __shared__ double part[32];
  if (threadIdx.x < 32) part[threadIdx.x] = 0.0;
  __syncthreads();
  const uint32_t nthreads = gridDim.x * blockDim.x;
  const uint32_t used = (nthreads / D) * D;             // stride is a multiple of D: fixed column
  const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)n_rows * D;
  if (id < used) {
    double s = 0.0;
    for (size_t e = id; e < total; e += used) {
      const float v = coords[e];
      if (fabsf(v) <= FLT_MAX) s += (double)v;          // non-finite entries do not poison the mean
    }
    atomicAdd(&part[id % D], s);
  }
  __syncthreads();
  if (threadIdx.x < D) atomicAdd(&sums[threadIdx.x], part[threadIdx.x]);
}

__global__ void image_kernel(const float* __restrict__ coords, uint32_t n_rows, uint32_t D,
                             uint32_t S, uint32_t T, const double* __restrict__ sums,
                             float* __restrict__ img, float* __restrict__ norms,
                             uint32_t* __restrict__ maxnorm_bits) {
  const uint32_t row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= 32 * T) return;
  const uint32_t t = row >> 5, c = row & 31;
  double nrm = 0.0;
  for (uint32_t k = 0; k < 2 * S; ++k) {
    float v = 0.0f;
    if (row < n_rows && k < D) {
      double mu = sums[k] / (double)n_rows;
      float muf = (float)mu;
      if (!(fabsf(muf) <= FLT_MAX)) muf = 0.0f;
      v = coords[(size_t)row * D + k] - muf;            // x' = fl(x - mu)
    }
    img[((size_t)t * S + (k >> 1)) * 64 + (k & 1) * 32 + c] = v;
    nrm += (double)v * (double)v;
  }
  // (a non-finite coordinate makes nrm non-finite, which raises the flag below)
  float nf = (row < n_rows) ? (float)nrm : INFINITY;    // pad rows can never be "inside"
  norms[row] = nf;
  if (row < n_rows) {
    if (nf <= kNormLimit)
      atomicMax(maxnorm_bits, __float_as_uint(nf));
    else
      atomicOr(maxnorm_bits + 1, 1u);   // NaN / inf / overflow-prone row: MFMA kernels stand down
  }
}

__global__ void fe_pad_kernel(const float* __restrict__ fe, uint32_t n_rows, uint32_t T,
                              float* __restrict__ out) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 32 * T) out[i] = (i < n_rows) ? fe[i] : INFINITY;
}

}  // namespace

#define DC_FOR_EACH_S(X) \
  X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16)
DC_FOR_EACH_S(DC_DECLARE_STEP)

bool mfma_supports(size_t n_cols) {
  if (n_cols < 1 || n_cols > 2 * (size_t)kMaxSteps) return false;
  return ((DC_STEP_MASK >> ((n_cols + 1) / 2 - 1)) & 1u) != 0;
}
size_t mfma_workspace_bytes(size_t n_rows, size_t n_cols) {
  if (!mfma_supports(n_cols) || n_rows == 0) return 0;
  return make_layout(n_rows, n_cols).total;
}

int mfma_prepare(const float* d_coords, uint32_t n_rows, uint32_t n_cols, void* d_ws,
                 hipStream_t stream) {
  const Layout L = make_layout(n_rows, n_cols);
  char* p = (char*)d_ws;
  if (hipMemsetAsync(p, 0, kHdrBytes, stream) != hipSuccess) return -1;
  const uint32_t blocks = (uint32_t)std::min<size_t>(512, ((size_t)n_rows * n_cols + 255) / 256);
  hipLaunchKernelGGL(colsum_kernel, dim3(blocks), dim3(256), 0, stream, d_coords, n_rows, n_cols,
                     (double*)(p + kHdrSums));
  hipLaunchKernelGGL(image_kernel, dim3((32 * L.T + 255) / 256), dim3(256), 0, stream, d_coords,
                     n_rows, n_cols, L.S, L.T, (const double*)(p + kHdrSums),
                     (float*)(p + L.off_img), (float*)(p + L.off_norm), (uint32_t*)p);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}

void launch_pop_mfma(const float* d_coords, uint32_t n_rows, uint32_t n_cols, uint32_t i_from,
                     uint32_t i_to, const Rad2& rad2, int n_rad, uint32_t* d_pops, void* d_ws,
                     hipStream_t stream) {
  switch ((n_cols + 1) / 2) {
#define X(SV)                                                                                   \
  case SV:                                                                                      \
    if ((DC_STEP_MASK >> (SV - 1)) & 1u)                                                        \
      pop_mfma_step_##SV(d_coords, n_rows, n_cols, d_ws, i_from, i_to, rad2, n_rad, d_pops, stream); \
    break;
    DC_FOR_EACH_S(X)
#undef X
    default:
      break;
  }
}

void launch_nn_mfma(const float* d_coords, uint32_t n_rows, uint32_t n_cols, const float* d_fe,
                    uint32_t i_from, uint32_t i_to, uint32_t* d_nn_idx, float* d_nn_d2,
                    uint32_t* d_hd_idx, float* d_hd_d2, void* d_ws, hipStream_t stream) {
  const Layout L = make_layout(n_rows, n_cols);
  hipLaunchKernelGGL(fe_pad_kernel, dim3((32 * L.T + 255) / 256), dim3(256), 0, stream, d_fe,
                     n_rows, L.T, (float*)((char*)d_ws + L.off_fe));
  switch ((n_cols + 1) / 2) {
#define X(SV)                                                                                \
  case SV:                                                                                   \
    if ((DC_STEP_MASK >> (SV - 1)) & 1u)                                                     \
      nn_mfma_step_##SV(d_coords, n_rows, n_cols, d_ws, i_from, i_to, d_nn_idx, d_nn_d2, d_hd_idx, \
                        d_hd_d2, stream);                                                    \
    break;
    DC_FOR_EACH_S(X)
#undef X
    default:
      break;
  }
}

}  // namespace dc
