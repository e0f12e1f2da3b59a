// semantics probe: v_permlane16_swap / v_permlane32_swap on gfx950 (prints, per lane, the two results for x = lane, y = 100 + lane)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* o) {
  const unsigned l = threadIdx.x;
  u2 a = __builtin_amdgcn_permlane16_swap(l, 100 + l, false, false);
  u2 b = __builtin_amdgcn_permlane32_swap(l, 100 + l, false, false);
  o[l * 4 + 0] = a.x; o[l * 4 + 1] = a.y; o[l * 4 + 2] = b.x; o[l * 4 + 3] = b.y;
}
int main() {
  unsigned *d, h[256];
  (void)hipMalloc(&d, sizeof h);
  k<<<1, 64>>>(d);
  (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; l += 4) printf("lane %2d: p16 (%3u, %3u)  p32 (%3u, %3u)\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
  return 0;
}
