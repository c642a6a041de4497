// Correspondences -> grid x grid cell index lists (integer work, bit-exact vs
// oracle/guidance_ref.py::cells_from_correspondences, which is pinned to the reference's
// GuidedStableDiffuser.process_correspondences, guided_stable_diffuser.py:490-584).
#include "common.h"
#include "compact.h"

namespace dh {

__global__ void k_valid(const long long* corr, int n, int img_res, uint8_t* valid) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  long long tx = corr[4 * (size_t)i + 2], ty = corr[4 * (size_t)i + 3];
  valid[i] = (tx >= 0 && tx < img_res && ty >= 0 && ty < img_res) ? 1 : 0;
}

__global__ void k_init_masks(uint8_t* m, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) m[i] = 1;
}

__global__ void k_pairs(const long long* corr, const int* idx, const int* count, int img_res, int grid, int* pairs,
                        uint8_t* bg_o, uint8_t* bg_t) {
  int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= *count) return;
  const long long* c = corr + 4 * (size_t)idx[r];
  const int cs = img_res / grid;
  int oc = (int)(c[1] / cs) * grid + (int)(c[0] / cs);
  int tc = (int)(c[3] / cs) * grid + (int)(c[2] / cs);
  pairs[2 * (size_t)r + 0] = oc;
  pairs[2 * (size_t)r + 1] = tc;
  bg_o[oc] = 0;
  bg_t[tc] = 0;
}

// scipy.ndimage.binary_erosion: cross structuring element, border_value 0, `iters` passes.
// blockIdx.x selects the mask (0 = orig, 1 = trans); one workgroup per mask, ping-pong in LDS.
__global__ void __launch_bounds__(1024) k_erode(uint8_t* masks, int grid, int iters) {
  extern __shared__ uint8_t sm[];
  const int n = grid * grid;
  uint8_t* m = masks + (size_t)blockIdx.x * n;
  uint8_t* a = sm;
  uint8_t* b = sm + n;
  for (int i = threadIdx.x; i < n; i += blockDim.x) a[i] = m[i];
  __syncthreads();
  for (int it = 0; it < iters; ++it) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
      int y = i / grid, x = i - y * grid;
      bool v = a[i] && y > 0 && y < grid - 1 && x > 0 && x < grid - 1;
      if (v) v = a[i - grid] && a[i + grid] && a[i - 1] && a[i + 1];
      b[i] = v ? 1 : 0;
    }
    __syncthreads();
    uint8_t* t = a; a = b; b = t;
  }
  for (int i = threadIdx.x; i < n; i += blockDim.x) m[i] = a[i];
}

// masks layout in: [orig][trans]; out layout [both][orig][trans]
__global__ void k_three_masks(const uint8_t* two, uint8_t* three, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  uint8_t o = two[i], t = two[n + i];
  three[i] = o & t;
  three[n + i] = o;
  three[2 * n + i] = t;
}

}  // namespace dh

using namespace dh;

extern "C" int dh_cells_workspace_bytes(int n, int grid, size_t* bytes) {
  DH_REQUIRE(n >= 0 && grid >= 1 && bytes, "bad arguments");
  size_t nn = n > 0 ? n : 1;
  *bytes = align_up(nn, 256) + align_up(nn * sizeof(int), 256) + align_up(2 * (size_t)grid * grid, 256) +
           (size_t)(3 * (cdiv((int)nn, CP_TILE) + cdiv(grid * grid, CP_TILE) + 2)) * sizeof(int) + 4096;
  return DH_OK;
}

extern "C" int dh_cells_from_correspondences(const int64_t* corr, int n, int img_res, int grid, int bg_erosion,
                                             int32_t* pairs, int32_t* bg_lists, uint8_t* bg_masks, int32_t* counts,
                                             void* workspace, size_t workspace_bytes, void* stream) {
  DH_REQUIRE(pairs && bg_lists && bg_masks && counts && workspace, "null pointer");
  DH_REQUIRE(n >= 0 && grid >= 1 && img_res >= grid && img_res % grid == 0, "bad sizes");
  DH_REQUIRE(2 * grid * grid <= 60000, "grid too large for the erosion kernel");
  size_t need;
  dh_cells_workspace_bytes(n, grid, &need);
  DH_REQUIRE(workspace_bytes >= need, "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int G2 = grid * grid;
  Arena a(workspace, workspace_bytes);
  uint8_t* valid = a.take<uint8_t>(n > 0 ? n : 1);
  int* idx = a.take<int>(n > 0 ? n : 1);
  uint8_t* two = a.take<uint8_t>(2 * (size_t)G2);
  int* bc = a.take<int>(3 * (cdiv(n > 0 ? n : 1, CP_TILE) + cdiv(G2, CP_TILE) + 2));
  DH_CHECK_HIP(hipMemsetAsync(counts, 0, 4 * sizeof(int), st));
  hipLaunchKernelGGL(k_init_masks, dim3(cdiv(2 * G2, 256)), dim3(256), 0, st, two, 2 * G2);
  if (n > 0) {
    DH_REQUIRE(corr, "null corr");
    hipLaunchKernelGGL(k_valid, dim3(cdiv(n, 256)), dim3(256), 0, st, (const long long*)corr, n, img_res, valid);
    compact(valid, n, 1, 0, idx, 0, counts, 1, bc, st);
    hipLaunchKernelGGL(k_pairs, dim3(cdiv(n, 256)), dim3(256), 0, st, (const long long*)corr, idx, counts, img_res,
                       grid, pairs, two, two + G2);
  }
  if (bg_erosion > 0)
    hipLaunchKernelGGL(k_erode, dim3(2), dim3(1024), 2 * G2, st, two, grid, bg_erosion);
  hipLaunchKernelGGL(k_three_masks, dim3(cdiv(G2, 256)), dim3(256), 0, st, two, bg_masks, G2);
  compact(bg_masks, G2, 3, G2, bg_lists, G2, counts + 1, 1, bc, st);
  DH_LAUNCH_CHECK();
  return DH_OK;
}
