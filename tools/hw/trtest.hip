// Hardware probe: semantics of ds_read_b64_tr_b16 on gfx950 (used by the bf16 filter-gradient kernel).
// Expected (guide T10): within each 16-lane group, lane 4q+p supplies the address of row q, columns 4p..4p+3;
// lane i receives column i of the 4 rows (row q in element q).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(const short* in, short* out) {
  __shared__ short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = in[i];
  __syncthreads();
  int lane = threadIdx.x;
  int q = (lane & 15) >> 2, p = lane & 3, g = lane >> 4;
  short* addr = &lds[(4 * g + q) * 32 + 4 * p];     // rows of 32 shorts; group g reads rows 4g..4g+3, cols 0..15
  v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)addr);
  for (int e = 0; e < 4; ++e) out[lane * 4 + e] = r[e];
}
int main() {
  short h[4096], o[256];
  for (int i = 0; i < 4096; ++i) h[i] = (short)i;
  short *di, *dout;
  hipMalloc(&di, sizeof(h)); hipMalloc(&dout, sizeof(o));
  hipMemcpy(di, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout);
  hipMemcpy(o, dout, sizeof(o), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int lane = 0; lane < 64; ++lane) {
    int g = lane >> 4, i = lane & 15;
    for (int e = 0; e < 4; ++e) {
      int expect = (4 * g + e) * 32 + i;
      if (o[lane * 4 + e] != expect) { if (bad < 8) printf("lane %d e %d got %d expect %d\n", lane, e, o[lane * 4 + e], expect); ++bad; }
    }
  }
  printf("trtest: %s (%d mismatches)\n", bad ? "MISMATCH" : "OK: lane i gets column i, element q = row q", bad);
  return bad != 0;
}
