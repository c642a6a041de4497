// depth_transform_mode = 'mesh' (depth_transform.py:91-195): the background depth and the rigidly moved
// foreground depth are triangulated per pixel quad (depth_to_mesh, :30-71), rasterised nearest-z with
// back-face culling, and the hit triangle's source image coordinates / foreground flag are read back
// (pytorch3d_renderer.py 'world_position' + 'flat_vertex_color' outputs, hard blend, one face per pixel).
//
// The reference draws with pytorch3d, which is not in the tree: parity is UNPINNED at this boundary.
// This TU restates pytorch3d's published rasterisation rule (rasterize_meshes, naive path):
//   * NDC with +X left / +Y up, pixel (col,row) centre at (1 - (2 col + 1)/R, 1 - (2 row + 1)/R)
//   * a face covers a pixel if the centre is inside it or within squared distance blur_radius (1e-5)
//   * barycentrics -> perspective correction -> clip to [0,1] + renormalise; z = sum b_i z_i, z < 0 skipped
//   * faces with |area| <= 1e-8 are skipped, area < 0 is a back face
//   * nearest z wins; exact ties go to the lower face index (background faces come first)
// in float32 with a fixed operation order and no FMA contraction (built with -ffp-contract=off), so that
// oracle/mesh_ref.py (NumPy float32, same order) reproduces the integer outputs bit for bit.
#include "common.h"
#include "compact.h"

namespace dh {

struct MeshXf {            // float32 Rodrigues parameters (transform_points, depth_transform.py:438-458)
  float ax, ay, az, c, s, tx, ty, tz, cx, cy, cz;
};

__device__ __forceinline__ float fdiv(float a, float b) { return (float)((double)a / (double)b); }

__device__ __forceinline__ void mesh_unproject(const float* depth, int p, int res, const float* gx, float invf, float& X,
                                               float& Y, float& Z) {
  const int row = p / res, col = p - row * res;
  const float d = depth[p];
  const float a = d * invf;
  X = -(a * gx[col]);
  Y = -(a * gx[row]);
  Z = d;
}

// vertex p of the background mesh / of the moved foreground mesh, already projected: (x_ndc, y_ndc, z)
__global__ void k_mesh_verts(const float* depth, const float* bg_depth, const uint8_t* mask, int res, const float* gx,
                             float invf, float f, MeshXf xf, float* vb, float* vf) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= res * res) return;
  float X, Y, Z;
  mesh_unproject(bg_depth, p, res, gx, invf, X, Y, Z);
  vb[3 * p + 0] = fdiv(f * X, Z);
  vb[3 * p + 1] = fdiv(f * Y, Z);
  vb[3 * p + 2] = Z;
  if (!mask[p]) return;
  mesh_unproject(depth, p, res, gx, invf, X, Y, Z);
  const float q0 = X - xf.cx, q1 = Y - xf.cy, q2 = Z - xf.cz;
  const float dot = (q0 * xf.ax + q1 * xf.ay) + q2 * xf.az;
  const float k1 = 1.f - xf.c;
  const float c0 = xf.ay * q2 - xf.az * q1, c1 = xf.az * q0 - xf.ax * q2, c2 = xf.ax * q1 - xf.ay * q0;
  const float r0 = (q0 * xf.c + c0 * xf.s) + (xf.ax * dot) * k1;
  const float r1 = (q1 * xf.c + c1 * xf.s) + (xf.ay * dot) * k1;
  const float r2 = (q2 * xf.c + c2 * xf.s) + (xf.az * dot) * k1;
  X = (r0 + xf.cx) + xf.tx;
  Y = (r1 + xf.cy) + xf.ty;
  Z = (r2 + xf.cz) + xf.tz;
  vf[3 * p + 0] = fdiv(f * X, Z);
  vf[3 * p + 1] = fdiv(f * Y, Z);
  vf[3 * p + 2] = Z;
}

struct Tri {
  float x0, y0, z0, x1, y1, z1, x2, y2, z2;
};

// face id -> its three vertices (false: the face does not exist / has an unmasked vertex)
__device__ __forceinline__ bool mesh_face(int fid, int res, const float* vb, const float* vf, const uint8_t* mask, Tri& t) {
  const int nq = (res - 1) * (res - 1);
  const bool fg = fid >= 2 * nq;
  const int loc = fg ? fid - 2 * nq : fid;
  const int q = loc >> 1, lower = loc & 1;
  const int y = q / (res - 1), x = q - y * (res - 1);
  const int v00 = y * res + x, v01 = v00 + 1, v10 = v00 + res, v11 = v10 + 1;
  const int i0 = v10, i1 = lower ? v11 : v01, i2 = lower ? v01 : v00;
  if (fg && !(mask[i0] && mask[i1] && mask[i2])) return false;
  const float* v = fg ? vf : vb;
  t.x0 = v[3 * i0]; t.y0 = v[3 * i0 + 1]; t.z0 = v[3 * i0 + 2];
  t.x1 = v[3 * i1]; t.y1 = v[3 * i1 + 1]; t.z1 = v[3 * i1 + 2];
  t.x2 = v[3 * i2]; t.y2 = v[3 * i2 + 1]; t.z2 = v[3 * i2 + 2];
  return true;
}

__device__ __forceinline__ float edge_fn(float px, float py, float ax, float ay, float bx, float by) {
  return (px - ax) * (by - ay) - (py - ay) * (bx - ax);
}

__device__ __forceinline__ float seg_dist2(float px, float py, float ax, float ay, float bx, float by) {
  const float dx = bx - ax, dy = by - ay;
  const float l2 = dx * dx + dy * dy;
  float t = 0.f;
  if (l2 > 1e-8f) {
    t = fdiv((px - ax) * dx + (py - ay) * dy, l2);
    t = t < 0.f ? 0.f : (t > 1.f ? 1.f : t);
  }
  const float qx = ax + t * dx, qy = ay + t * dy;
  return (px - qx) * (px - qx) + (py - qy) * (py - qy);
}

constexpr float MESH_EPS = 1e-8f;

// coverage + clipped perspective-correct barycentrics of pixel centre (px, py); returns false if not drawn
__device__ __forceinline__ bool mesh_hit(const Tri& t, float px, float py, float blur, float& b0, float& b1, float& b2,
                                         float& pz) {
  const float area = edge_fn(t.x2, t.y2, t.x0, t.y0, t.x1, t.y1);
  if (area < 0.f) return false;                                 // back face
  if (area <= MESH_EPS) return false;                            // degenerate
  const float w0 = fdiv(edge_fn(px, py, t.x1, t.y1, t.x2, t.y2), area);
  const float w1 = fdiv(edge_fn(px, py, t.x2, t.y2, t.x0, t.y0), area);
  const float w2 = fdiv(edge_fn(px, py, t.x0, t.y0, t.x1, t.y1), area);
  const bool inside = w0 > 0.f && w1 > 0.f && w2 > 0.f;
  if (!inside) {
    float d = seg_dist2(px, py, t.x0, t.y0, t.x1, t.y1);
    const float d1 = seg_dist2(px, py, t.x1, t.y1, t.x2, t.y2);
    const float d2 = seg_dist2(px, py, t.x2, t.y2, t.x0, t.y0);
    d = d1 < d ? d1 : d;
    d = d2 < d ? d2 : d;
    if (d > blur) return false;
  }
  // perspective correction
  const float t0 = (w0 * t.z1) * t.z2, t1 = (t.z0 * w1) * t.z2, t2 = (t.z0 * t.z1) * w2;
  float den = (t0 + t1) + t2;
  den = den > MESH_EPS ? den : MESH_EPS;
  float p0 = fdiv(t0, den), p1 = fdiv(t1, den), p2 = fdiv(t2, den);
  // clip + renormalise
  p0 = p0 < 0.f ? 0.f : (p0 > 1.f ? 1.f : p0);
  p1 = p1 < 0.f ? 0.f : (p1 > 1.f ? 1.f : p1);
  p2 = p2 < 0.f ? 0.f : (p2 > 1.f ? 1.f : p2);
  float sum = (p0 + p1) + p2;
  sum = sum > MESH_EPS ? sum : MESH_EPS;
  b0 = fdiv(p0, sum); b1 = fdiv(p1, sum); b2 = fdiv(p2, sum);
  pz = (b0 * t.z0 + b1 * t.z1) + b2 * t.z2;
  return pz >= 0.f;
}

__device__ __forceinline__ float pix_centre(int i, int res) { return 1.f - fdiv((float)(2 * i + 1), (float)res); }

// one thread per face: every covered pixel of its (blur-expanded) bounding box bids (z bits, face id)
__global__ void k_mesh_raster(int res, const float* vb, const float* vf, const uint8_t* mask, float blur, float pad,
                              unsigned long long* zbuf) {
  const int nq = (res - 1) * (res - 1);
  const int fid = blockIdx.x * blockDim.x + threadIdx.x;
  if (fid >= 4 * nq) return;
  Tri t;
  if (!mesh_face(fid, res, vb, vf, mask, t)) return;
  if (!(t.z0 > 0.f && t.z1 > 0.f && t.z2 > 0.f)) return;          // a vertex behind the camera: dropped (no clipping)
  float xmin = fminf(t.x0, fminf(t.x1, t.x2)) - pad, xmax = fmaxf(t.x0, fmaxf(t.x1, t.x2)) + pad;
  float ymin = fminf(t.y0, fminf(t.y1, t.y2)) - pad, ymax = fmaxf(t.y0, fmaxf(t.y1, t.y2)) + pad;
  // pixel centre x = 1 - (2 col + 1) / R  <=>  col = ((1 - x) R - 1) / 2 ; a conservative integer range
  int c_lo = (int)floorf(((1.f - xmax) * (float)res - 1.f) * 0.5f) - 1, c_hi = (int)ceilf(((1.f - xmin) * (float)res - 1.f) * 0.5f) + 1;
  int r_lo = (int)floorf(((1.f - ymax) * (float)res - 1.f) * 0.5f) - 1, r_hi = (int)ceilf(((1.f - ymin) * (float)res - 1.f) * 0.5f) + 1;
  c_lo = c_lo < 0 ? 0 : c_lo; r_lo = r_lo < 0 ? 0 : r_lo;
  c_hi = c_hi > res - 1 ? res - 1 : c_hi; r_hi = r_hi > res - 1 ? res - 1 : r_hi;
  for (int r = r_lo; r <= r_hi; ++r) {
    const float py = pix_centre(r, res);
    for (int c = c_lo; c <= c_hi; ++c) {
      const float px = pix_centre(c, res);
      float b0, b1, b2, pz;
      if (!mesh_hit(t, px, py, blur, b0, b1, b2, pz)) continue;
      const unsigned long long key = ((unsigned long long)__float_as_uint(pz) << 32) | (unsigned)fid;
      atomicMin(&zbuf[(size_t)r * res + c], key);
    }
  }
}

// per pixel: winner -> depth, interpolated source coordinates (rounded half-even), foreground flag
__global__ void k_mesh_resolve(int res, const float* vb, const float* vf, const uint8_t* mask, const float* lin01, float blur,
                               const unsigned long long* zbuf, float* zmap, int* src_xy, uint8_t* fgflag,
                               unsigned int* minmax) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= res * res) return;
  const unsigned long long key = zbuf[p];
  float z = 0.f;                                           // background colour of an uncovered pixel
  int sx = 0, sy = 0;
  uint8_t fg = 0;
  if (key != ~0ull) {
    const int fid = (int)(unsigned)(key & 0xffffffffull);
    const int nq = (res - 1) * (res - 1);
    Tri t;
    mesh_face(fid, res, vb, vf, mask, t);
    const int row = p / res, col = p - row * res;
    float b0, b1, b2, pz;
    mesh_hit(t, pix_centre(col, res), pix_centre(row, res), blur, b0, b1, b2, pz);
    z = pz;
    const bool isfg = fid >= 2 * nq;
    const int loc = isfg ? fid - 2 * nq : fid;
    const int q = loc >> 1, lower = loc & 1;
    const int y = q / (res - 1), x = q - y * (res - 1);
    // vertex order of the face: (y+1,x), then (y+1,x+1)|(y,x+1), then (y,x+1)|(y,x)
    const int x0 = x, y0 = y + 1, x1 = x + 1, y1 = lower ? y + 1 : y, x2 = lower ? x + 1 : x, y2 = y;
    const float u = (b0 * lin01[x0] + b1 * lin01[x1]) + b2 * lin01[x2];
    const float v = (b0 * lin01[y0] + b1 * lin01[y1]) + b2 * lin01[y2];
    sx = (int)rintf(u * (float)(res - 1));
    sy = (int)rintf(v * (float)(res - 1));
    fg = isfg ? 1 : 0;
  }
  zmap[p] = z;
  src_xy[2 * p] = sx;
  src_xy[2 * p + 1] = sy;
  fgflag[p] = fg;
  // disparity bounds (1/z > 0 for covered pixels: uint order = float order)
  const float dsp = fdiv(1.f, z);
  atomicMin(&minmax[0], __float_as_uint(dsp));
  atomicMax(&minmax[1], __float_as_uint(dsp));
}

__global__ void k_mesh_finish(int res, const float* zmap, const unsigned int* minmax, const float* bounds, const int* fg_list,
                              const int* count, const int* src_xy, float* disparity, long long* corr) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  const int R2 = res * res;
  if (p < R2) {
    const float lo = bounds ? bounds[0] : __uint_as_float(minmax[0]);
    const float hi = bounds ? bounds[1] : __uint_as_float(minmax[1]);
    const float dsp = fdiv(1.f, zmap[p]);
    disparity[p] = fdiv(255.f * (dsp - lo), hi - lo);
  }
  if (p < count[0]) {
    const int pix = fg_list[p];
    corr[4 * (size_t)p + 0] = src_xy[2 * pix];
    corr[4 * (size_t)p + 1] = src_xy[2 * pix + 1];
    corr[4 * (size_t)p + 2] = pix % res;
    corr[4 * (size_t)p + 3] = pix / res;
  }
}

}  // namespace dh

using namespace dh;

extern "C" int dh_mesh_workspace_bytes(int res, size_t* bytes) {
  DH_REQUIRE(res >= 2 && bytes, "bad arguments");
  const size_t R2 = (size_t)res * res;
  *bytes = 2 * align_up(R2 * 12, 256) + align_up(R2 * 8, 256) + align_up(R2 * 8, 256) + align_up(R2, 256) +
           align_up(R2 * 4, 256) + (size_t)(cdiv((int)R2, CP_TILE) + 2) * 4 + 4096;
  return DH_OK;
}

// xform: 11 floats {axis (unit) x3, cos, sin, translation x3, centroid x3}; bounds: NULL or {lo, hi} of the input
// disparity.  Outputs: zmap [res^2] f32 (rendered depth), disparity [res^2] f32, fg_flag [res^2] u8,
// corr [<= res^2][4] i64 (src_x, src_y, tgt_x, tgt_y) in row-major target order, counts[0] = number of rows.
extern "C" int dh_mesh_reproject(const float* depth, const float* bg_depth, const uint8_t* fg_mask, int res,
                                 const float* grid, const float* lin01, float inv_f, float f, const float* xform,
                                 const float* bounds, float blur_radius, float* zmap, float* disparity, uint8_t* fg_flag,
                                 int64_t* corr, int32_t* counts, void* workspace, size_t workspace_bytes, void* stream) {
  DH_REQUIRE(depth && bg_depth && fg_mask && grid && lin01 && xform, "null input");
  DH_REQUIRE(zmap && disparity && fg_flag && corr && counts && workspace, "null output");
  DH_REQUIRE(res >= 2 && blur_radius >= 0.f, "bad arguments");
  size_t need;
  dh_mesh_workspace_bytes(res, &need);
  DH_REQUIRE(workspace_bytes >= need, "workspace too small");
  hipStream_t st = (hipStream_t)stream;
  const int R2 = res * res;
  Arena a(workspace, workspace_bytes);
  float* vb = a.take<float>((size_t)3 * R2);
  float* vf = a.take<float>((size_t)3 * R2);
  unsigned long long* zbuf = a.take<unsigned long long>(R2);
  int* src_xy = a.take<int>((size_t)2 * R2);
  int* fg_list = a.take<int>(R2);
  int* bc = a.take<int>(cdiv(R2, CP_TILE) + 2);
  unsigned int* minmax = a.take<unsigned int>(2);
  MeshXf xf;
  xf.ax = xform[0]; xf.ay = xform[1]; xf.az = xform[2]; xf.c = xform[3]; xf.s = xform[4];
  xf.tx = xform[5]; xf.ty = xform[6]; xf.tz = xform[7]; xf.cx = xform[8]; xf.cy = xform[9]; xf.cz = xform[10];
  DH_CHECK_HIP(hipMemsetAsync(zbuf, 0xff, (size_t)R2 * 8, st));
  DH_CHECK_HIP(hipMemsetAsync(minmax, 0xff, 4, st));
  DH_CHECK_HIP(hipMemsetAsync(minmax + 1, 0, 4, st));
  DH_CHECK_HIP(hipMemsetAsync(counts, 0, 4 * sizeof(int), st));
  hipLaunchKernelGGL(k_mesh_verts, dim3(cdiv(R2, 256)), dim3(256), 0, st, depth, bg_depth, fg_mask, res, grid, inv_f, f, xf,
                     vb, vf);
  const int nfaces = 4 * (res - 1) * (res - 1);
  const float pad = sqrtf(blur_radius) + 1e-6f;
  hipLaunchKernelGGL(k_mesh_raster, dim3(cdiv(nfaces, 256)), dim3(256), 0, st, res, vb, vf, fg_mask, blur_radius, pad, zbuf);
  hipLaunchKernelGGL(k_mesh_resolve, dim3(cdiv(R2, 256)), dim3(256), 0, st, res, vb, vf, fg_mask, lin01, blur_radius, zbuf,
                     zmap, src_xy, fg_flag, minmax);
  compact(fg_flag, R2, 1, 0, fg_list, 0, counts, 1, bc, st);
  hipLaunchKernelGGL(k_mesh_finish, dim3(cdiv(R2, 256)), dim3(256), 0, st, res, zmap, minmax, bounds, fg_list, counts, src_xy,
                     disparity, (long long*)corr);
  DH_LAUNCH_CHECK();
  return DH_OK;
}
