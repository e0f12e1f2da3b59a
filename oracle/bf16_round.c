/* bf16 rounding for the ORACLE (test infrastructure only, never linked into the product libraries).
 *
 * float32 -> nearest-even bf16 -> float32, the arithmetic of __float2bfloat16 that oracle/bf16.py states in numpy
 * (bias 0x7FFF + lsb of the kept half, NaN -> 0x7FC0): ONE pass, a fixed number of OpenMP threads.  A full-depth oracle pass
 * rounds ~3 G activation elements; numpy needs five passes over them on one thread (~45 s of a 125 s pass on the pool's hosts).
 * tests/test_oracle_kats.py pins this helper against the numpy statement on random bit patterns. */
#include <stddef.h>
#include <stdint.h>

void pegainfer_oracle_bf16_round(const uint32_t* in, uint32_t* out, size_t n, int threads) {
#pragma omp parallel for schedule(static) num_threads(threads)
  for (size_t i = 0; i < n; ++i) {
    const uint32_t u = in[i];
    out[i] = ((u & 0x7FFFFFFFu) > 0x7F800000u) ? 0x7FC00000u : ((u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u);
  }
}
