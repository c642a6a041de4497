// probe of ds_read_b64_tr_b16 semantics on gfx950: which elements does each lane receive?
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(unsigned short* out, int stride) {
  __shared__ __attribute__((aligned(16))) unsigned short sm[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) sm[i] = (unsigned short)i;
  __syncthreads();
  const int l = threadIdx.x, t = l & 15, g = l >> 4;
  // lane t of group g supplies the 8-byte piece (row t/4, cols 4*(t%4)..) of a 4 x 16 block whose rows are `stride` halves apart
  const int addr = g * 1024 + (t >> 2) * stride + 4 * (t & 3);
  v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(sm + addr));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = r[j];
}
int main() {
  unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
  unsigned short h[256];
  for (int stride : {16, 72}) {
    k<<<1, 64>>>(d, stride); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("stride %d\n", stride);
    for (int l = 0; l < 64; l += 1) if (l < 6 || (l % 16) == 15 || l == 16 || l == 33) printf(" lane %2d: %5d %5d %5d %5d\n", l, h[4*l], h[4*l+1], h[4*l+2], h[4*l+3]);
  }
  return 0;
}
