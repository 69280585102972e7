// Probe of the operand / result layout of v_mfma_f64_4x4x4_4b_f64 on gfx950: for every pair of
// lanes (la, lb) A = 1 in lane la, B = 1 in lane lb, 0 elsewhere; prints which result lane is 1.
//   hipcc --offload-arch=gfx950 -O2 tools/mfma_probe.hip -o tools/_stamps/mfma_probe ; gpurun -- tools/_stamps/mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void probe(int* out) {
  const int lane = threadIdx.x;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb) {
      const double a = lane == la ? 1. : 0., b = lane == lb ? 1. : 0.;
      const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0., 0, 0, 0);
      const unsigned long long m = __ballot(d != 0.);
      if (lane == 0) out[la * 64 + lb] = m == 0ull ? -1 : (__popcll(m) == 1 ? __builtin_ctzll(m) : -2);
    }
}

int main() {
  int* d;
  hipMalloc(&d, 64 * 64 * sizeof(int));
  probe<<<1, 64>>>(d);
  static int h[64 * 64];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int la = 0; la < 64; ++la) {
    printf("A lane %2d:", la);
    for (int lb = 0; lb < 64; ++lb)
      if (h[la * 64 + lb] != -1) printf(" B%d->D%d", lb, h[la * 64 + lb]);
    printf("\n");
  }
  return 0;
}
