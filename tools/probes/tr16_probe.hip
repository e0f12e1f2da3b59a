#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(const uint16_t* in, uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = in[i];
  __syncthreads();
  const int l = threadIdx.x;
  // lane address: contiguous 8 bytes per lane
  v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)(lds + l * 4));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (uint16_t)r[j];
}
int main() {
  uint16_t h[4096], o[256];
  for (int i = 0; i < 4096; ++i) h[i] = i;
  uint16_t *di, *dout;
  hipMalloc(&di, sizeof h); hipMalloc(&dout, sizeof o);
  hipMemcpy(di, h, sizeof h, hipMemcpyHostToDevice);
  k<<<1, 64>>>(di, dout);
  hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" %4d", o[l * 4 + j]); printf("\n"); }
  return 0;
}
