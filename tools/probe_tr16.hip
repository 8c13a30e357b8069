// Probe of gfx950's ds_read_b64_tr_b16: lane l addresses elements 4 l .. 4 l + 3 (value = its own index) and prints what
// it receives.  The documented behaviour (csrc/mf_common.h): lane c of a 16-lane group gets, as element r, element
// (c & 3) of lane 4 r + (c >> 2) of the group.   hipcc --offload-arch=gfx950 -O2 -o /tmp/probe_tr16 tools/probe_tr16.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4i16 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned short *out) {
  __shared__ __attribute__((aligned(16))) unsigned short s[256];
  for (int e = 0; e < 4; ++e) s[4 * threadIdx.x + e] = 4 * threadIdx.x + e;
  __syncthreads();
  v4i16 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16 *)(s + 4 * threadIdx.x));
  out[4 * threadIdx.x + 0] = v.x; out[4 * threadIdx.x + 1] = v.y; out[4 * threadIdx.x + 2] = v.z; out[4 * threadIdx.x + 3] = v.w;
}
int main() {
  unsigned short *d, h[256];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    const int base = l & ~15, c = l & 15;
    printf("lane %2d:", l);
    for (int r = 0; r < 4; ++r) {
      const int want = 4 * (base + 4 * r + (c >> 2)) + (c & 3);
      printf(" %3d%s", h[4 * l + r], h[4 * l + r] == want ? "" : "!");
      bad += h[4 * l + r] != want;
    }
    printf("\n");
  }
  printf("mismatches vs documented behaviour: %d\n", bad);
  return bad != 0;
}
