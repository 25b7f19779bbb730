#include <hip/hip_runtime.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
__global__ void k(const unsigned short* in, unsigned short* out, const int* addr) {
  extern __shared__ __attribute__((aligned(16))) unsigned short sm[];
  for (int i = threadIdx.x; i < 4096; i += 64) sm[i] = in[i];
  __syncthreads();
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)((__attribute__((address_space(3))) char*)sm + addr[threadIdx.x]));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (unsigned short)v[j];
}
int main() {
  unsigned short h[4096]; for (int i = 0; i < 4096; ++i) h[i] = i;
  unsigned short *din, *dout; int* da; int ha[64];
  // lane p of each 16-lane group g: piece p -> row p/4, cols 4*(p%4); rows of 64 elements (128 B pitch); group g uses rows 4g..4g+3
  for (int l = 0; l < 64; ++l) { int g = l >> 4, p = l & 15; ha[l] = ((4 * g + (p >> 2)) * 64 + 4 * (p & 3)) * 2; }
  hipMalloc(&din, 8192); hipMalloc(&dout, 512); hipMalloc(&da, 256);
  hipMemcpy(din, h, 8192, hipMemcpyHostToDevice); hipMemcpy(da, ha, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 8192, 0, din, dout, da);
  unsigned short o[256]; hipMemcpy(o, dout, 512, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d   (row,col of each: %d,%d %d,%d %d,%d %d,%d)\n", l, o[4*l], o[4*l+1], o[4*l+2], o[4*l+3],
     o[4*l]/64, o[4*l]%64, o[4*l+1]/64, o[4*l+1]%64, o[4*l+2]/64, o[4*l+2]%64, o[4*l+3]/64, o[4*l+3]%64);
  return 0;
}
