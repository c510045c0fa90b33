// Lane map of ds_read_b64_tr_b16 (gfx950) as csrc/decoder_bwd.hip k_wgrad6 uses it: a [rows][64 halves] LDS image, group g of 16 lanes
// reads rows 8g .. 8g+3; lane 4q+p of the group supplies the address of row q, columns 4p .. 4p+3.  Expected (cdna_hip_programming.md T10):
// lane i of the group receives column i of the four rows.   hipcc --offload-arch=gfx950 -O3 -o trread trread.hip && ./trread
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4 __attribute__((__vector_size__(4 * sizeof(short))));
__global__ void k(const float* in, float* out) {
  __shared__ __attribute__((aligned(16))) _Float16 lds[64 * 64];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (_Float16)in[i];
  __syncthreads();
  const int lane = threadIdx.x, g = lane >> 4, i = lane & 15, q = i >> 2, p = i & 3;
  const _Float16* a = lds + (8 * g + q) * 64 + 4 * p;
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)a);
  for (int c = 0; c < 4; ++c) { _Float16 h; short s = v[c]; __builtin_memcpy(&h, &s, 2); out[lane * 4 + c] = (float)h; }
}
int main() {
  float h_in[4096], h_out[256];
  for (int r = 0; r < 64; ++r) for (int c = 0; c < 64; ++c) h_in[r * 64 + c] = float((r % 32) * 64 + c);
  float *d_in, *d_out;
  hipMalloc(&d_in, sizeof(h_in)); hipMalloc(&d_out, sizeof(h_out));
  hipMemcpy(d_in, h_in, sizeof(h_in), hipMemcpyHostToDevice);
  k<<<1, 64>>>(d_in, d_out);
  hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int lane = 0; lane < 64; ++lane)
    for (int c = 0; c < 4; ++c) {
      const int g = lane >> 4, i = lane & 15;
      const float want = float((8 * g + c) * 64 + i);
      if (h_out[lane * 4 + c] != want) { if (bad < 8) printf("lane %d elem %d: got %.0f (row %d col %d) want %.0f\n", lane, c, h_out[lane * 4 + c], int(h_out[lane * 4 + c]) / 64, int(h_out[lane * 4 + c]) % 64, want); ++bad; }
    }
  printf("tr read lane map: %s (%d mismatches)\n", bad ? "DIFFERENT" : "as expected", bad);
  return bad != 0;
}
