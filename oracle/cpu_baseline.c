/* cpu_baseline.c - the CPU row next to the GPU numbers (bench.py "cpu_baseline", kind "port").
 *
 * TEST / MEASUREMENT INFRASTRUCTURE, never linked into the product. libxsmm itself is not in the image
 * (cmake/modules/xsmm.cmake:14-18 fetches it at configure time), so this restates the CALL STRUCTURE the
 * reference runs on a CPU for BASELINE config C2 instead of timing libxsmm:
 *   - operands in the compiler's packed block layouts (ToBlockLayoutAndBack.cpp:460-471, default 32x32x32):
 *     A [MB][KB][32][32] (m, k), B [NB][KB][32][32] (k, n), C [MB][NB][32][32];
 *   - an OpenMP loop over the (MB, NB) tile grid (omp.wsloop over scf.parallel,
 *     test/Passes/pass-convert-mlp-to-parallel-tile.mlir:80-88; DefaultPipeline.cpp:179-180);
 *   - per tile ONE batch-reduce call: C_tile (+)= sum_kb A[mb][kb] * B[nb][kb] with the brgemm dispatch
 *     [32,32,32,32,32,32,1024,1024] (pass-convert-gemm-to-parallel-tile.mlir:29), here a register-blocked
 *     32x32x32 microkernel (4 rows x 16 columns of accumulators, k-ordered fma chain per element like the
 *     oracle) that gcc vectorises for the host it is compiled on.
 * Build: gcc -O3 -march=native -fopenmp (bench.py compiles it on the machine it runs on; the Makefile builds
 * a portable x86-64-v3 fallback). Checked against oracle/xsmm_oracle.c by tests/test_oracle_golden.py. */
#include <stdint.h>
#include <string.h>

#define TB 32

/* row-major [rows][cols] (ld) -> blocks [rows/32][cols/32][32][32] */
void cpu_pack_blocks(const float *src, int64_t rows, int64_t cols, int64_t ld, float *dst) {
  const int64_t rb = rows / TB, cb = cols / TB;
  for (int64_t r = 0; r < rb; ++r)
    for (int64_t c = 0; c < cb; ++c)
      for (int i = 0; i < TB; ++i)
        memcpy(dst + ((r * cb + c) * TB + i) * TB, src + (r * TB + i) * ld + c * TB, TB * sizeof(float));
}

void cpu_unpack_blocks(const float *src, int64_t rows, int64_t cols, int64_t ld, float *dst) {
  const int64_t rb = rows / TB, cb = cols / TB;
  for (int64_t r = 0; r < rb; ++r)
    for (int64_t c = 0; c < cb; ++c)
      for (int i = 0; i < TB; ++i)
        memcpy(dst + (r * TB + i) * ld + c * TB, src + ((r * cb + c) * TB + i) * TB, TB * sizeof(float));
}

/* B is stored [k/32][n/32][32][32] by cpu_pack_blocks; the tile loop wants [NB][KB]: transpose the block grid */
void cpu_pack_b(const float *src, int64_t k, int64_t n, int64_t ld, float *dst) {
  const int64_t kb = k / TB, nb = n / TB;
  for (int64_t c = 0; c < nb; ++c)
    for (int64_t r = 0; r < kb; ++r)
      for (int i = 0; i < TB; ++i)
        memcpy(dst + ((c * kb + r) * TB + i) * TB, src + (r * TB + i) * ld + c * TB, TB * sizeof(float));
}

/* one brgemm invoke of the reference's tile dispatch: C[32][32] (+)= sum_{b<br} A_b[32][32] * B_b[32][32].
 * Register block RB rows x 2 vectors of columns held in vector registers over the whole batch-reduce loop
 * (AVX-512: 8 x 32 columns = 16 zmm; otherwise 4 x 16 columns = 8 ymm), one fma chain per element in k order. */
#if defined(__AVX512F__)
#define VL 16
#define RB 8
#else
#define VL 8
#define RB 4
#endif
typedef float vf __attribute__((vector_size(VL * 4), aligned(4)));

static inline void brgemm_tile(const float *restrict A, const float *restrict B, float *restrict C, int64_t br,
                               int beta0) {
  for (int i0 = 0; i0 < TB; i0 += RB)
    for (int j0 = 0; j0 < TB; j0 += 2 * VL) {
      vf acc[RB][2];
      for (int r = 0; r < RB; ++r)
        for (int h = 0; h < 2; ++h) {
          if (beta0) acc[r][h] = (vf){0};
          else acc[r][h] = *(const vf *)(C + (i0 + r) * TB + j0 + h * VL);
        }
      for (int64_t b = 0; b < br; ++b) {
        const float *a = A + b * TB * TB + i0 * TB, *bb = B + b * TB * TB + j0;
        for (int kk = 0; kk < TB; ++kk) {
          const vf b0 = *(const vf *)(bb + kk * TB), b1 = *(const vf *)(bb + kk * TB + VL);
#pragma GCC unroll 8
          for (int r = 0; r < RB; ++r) {
            const float av = a[r * TB + kk];
            acc[r][0] += av * b0;
            acc[r][1] += av * b1;
          }
        }
      }
      for (int r = 0; r < RB; ++r)
        for (int h = 0; h < 2; ++h) *(vf *)(C + (i0 + r) * TB + j0 + h * VL) = acc[r][h];
    }
}

/* `reps` passes of C = (beta0 ? 0 : C) + A * B over packed operands; m, n, k multiples of 32 */
int cpu_brgemm_tiled_f32(int64_t m, int64_t n, int64_t k, const float *Ap, const float *Bp, float *Cp, int beta0,
                         int64_t reps) {
  if (m % TB || n % TB || k % TB) return 1;
  const int64_t mb = m / TB, nb = n / TB, kb = k / TB;
  for (int64_t it = 0; it < reps; ++it) {
#pragma omp parallel for collapse(2) schedule(static)
    for (int64_t i = 0; i < mb; ++i)
      for (int64_t j = 0; j < nb; ++j)
        brgemm_tile(Ap + i * kb * TB * TB, Bp + j * kb * TB * TB, Cp + (i * nb + j) * TB * TB, kb, beta0);
  }
  return 0;
}

int cpu_baseline_threads(void) {
  extern int omp_get_max_threads(void);
  return omp_get_max_threads();
}
/* threads of the following runs (bench.py sizes the team to the CPUs the container may really use: affinity mask AND cgroup quota -
 * a 128-thread team under a 16-CPU quota is throttled to a fraction of what 16 threads deliver) */
void cpu_baseline_set_threads(int n) {
  extern void omp_set_num_threads(int);
  if (n > 0) omp_set_num_threads(n);
}
