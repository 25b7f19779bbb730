/* driver.c - stands where tpp-run stands for tests/abi/xsmm_calls.ll: allocates the "memrefs" (64-byte aligned host memory, as
 * memref.alloc does: test/Passes/DefaultPipeline/default-tpp-passes.mlir:106), fills them from a file, calls one entry function
 * of the LLVM-IR module, writes the buffers back. It knows nothing of the runtime's header - only the entry points of the module.
 *   driver <entry> <in.bin> <out.bin> <bytes of buffer 0> [<bytes of buffer 1> ...]     (buffers are concatenated in the files) */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

void fusion_f32(void *, void *, void *, void *);
void quarternary_bf16_amx(void *, void *, void *, void *);
void brgemm_bf16_amx(void *, void *, void *);
void gemm_bf16(void *, void *, void *);
void zero_f32(void *);
void binary_add_f32(void *, void *, void *);
double fill_scalar_f32_timed(void *, float);

int main(int argc, char **argv) {
  if (argc < 5) return 2;
  const char *entry = argv[1];
  int nb = argc - 4;
  void *buf[4] = {0, 0, 0, 0};
  size_t sz[4] = {0, 0, 0, 0};
  if (nb > 4) return 2;
  FILE *f = fopen(argv[2], "rb");
  if (!f) return 3;
  for (int i = 0; i < nb; ++i) {
    sz[i] = (size_t)atol(argv[4 + i]);
    if (posix_memalign(&buf[i], 64, sz[i] ? sz[i] : 64)) return 4;
    if (fread(buf[i], 1, sz[i], f) != sz[i]) return 5;
  }
  fclose(f);
  if (!strcmp(entry, "fusion_f32") && nb == 4) fusion_f32(buf[0], buf[1], buf[2], buf[3]);
  else if (!strcmp(entry, "quarternary_bf16_amx") && nb == 4) quarternary_bf16_amx(buf[0], buf[1], buf[2], buf[3]);
  else if (!strcmp(entry, "brgemm_bf16_amx") && nb == 3) brgemm_bf16_amx(buf[0], buf[1], buf[2]);
  else if (!strcmp(entry, "gemm_bf16") && nb == 3) gemm_bf16(buf[0], buf[1], buf[2]);
  else if (!strcmp(entry, "zero_f32") && nb == 1) zero_f32(buf[0]);
  else if (!strcmp(entry, "binary_add_f32") && nb == 3) binary_add_f32(buf[0], buf[1], buf[2]);
  else if (!strcmp(entry, "fill_scalar_f32_timed") && nb == 1) {
    const double s = fill_scalar_f32_timed(buf[0], 7.5f);
    printf("seconds %.9f\n", s);
    if (!(s >= 0.0 && s < 60.0)) return 7;
  } else return 6;
  f = fopen(argv[3], "wb");
  if (!f) return 3;
  for (int i = 0; i < nb; ++i) fwrite(buf[i], 1, sz[i], f);
  fclose(f);
  return 0;
}
