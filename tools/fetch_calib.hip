// fetch_calib.hip -- what does rocprofv3's FETCH_SIZE count for the access shapes of this engine?
// Three kernels over the same 64 MiB buffer of bytes, each touching a KNOWN set of 128-byte lines
// exactly once (the buffer is far larger than L2 + MALL reuse matters: one pass, cold):
//   stream16   every lane loads 16 contiguous bytes (frame_max_kernel's shape): all bytes
//   line1      every lane loads ONE byte of a 128-byte line of its own: 1/128 of the bytes, all lines
//   window13   every wave loads the 13 x 13 window of a cluster like the refine kernels: 13
//              consecutive bytes of 13 rows 512 bytes apart, windows on a 16 x 16 grid: 13 of
//              every 16 image rows, all four lines of such a row (eight windows share a line)
// Build and run on the GPU box (tools/profile_r3.sh):
//   hipcc --offload-arch=gfx950 -O2 tools/fetch_calib.hip -o /tmp/fetch_calib
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -- /tmp/fetch_calib
// The per-kernel counter values against the known line counts calibrate `roofline.traffic`.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

constexpr size_t N = 64ull << 20;

__global__ void stream16(const uint4* __restrict__ p, unsigned* sink, size_t n16) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned acc = 0;
  if (i < n16) { const uint4 v = p[i]; acc = v.x ^ v.y ^ v.z ^ v.w; }
  if (acc == 0x12345678u) *sink = acc;
}

__global__ void line1(const unsigned char* __restrict__ p, unsigned* sink, size_t nlines) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned acc = 0;
  if (i < nlines) acc = p[i * 128 + (i % 128)];
  if (acc == 77u) *sink = acc;   // (never true: the buffer holds 1s; keeps the load)
}

// frames of 512 x 512 bytes; wave w takes window w: origin (16 * (w / 32) % 496, 16 * (w % 32)) of
// frame w / (31 * 32); lane l < 169 loads byte (l / 13, l % 13) of it
__global__ void window13(const unsigned char* __restrict__ p, unsigned* sink, size_t nwin) {
  const size_t w = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / 192;   // 3 waves of 64 lanes per window
  const int l = (int)(((size_t)blockIdx.x * blockDim.x + threadIdx.x) % 192);
  unsigned acc = 0;
  if (w < nwin && l < 169) {
    const size_t frame = w / (31 * 32), r = w % (31 * 32);
    const size_t y0 = 16 * (r / 32), x0 = 16 * (r % 32);
    acc = p[frame * 512 * 512 + (y0 + l / 13) * 512 + x0 + l % 13];
  }
  if (acc == 77u) *sink = acc;   // (never true: the buffer holds 1s; keeps the load)
}

int main() {
  unsigned char* buf = nullptr;
  unsigned* sink = nullptr;
  if (hipMalloc(&buf, N) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return 1;
  if (hipMemset(buf, 1, N) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return 1;
  const size_t n16 = N / 16, nlines = N / 128, nwin = (N / (512 * 512)) * 31 * 32;
  for (int rep = 0; rep < 3; ++rep) {
    stream16<<<dim3((unsigned)((n16 + 255) / 256)), dim3(256)>>>((const uint4*)buf, sink, n16);
    (void)hipDeviceSynchronize();
    line1<<<dim3((unsigned)((nlines + 255) / 256)), dim3(256)>>>(buf, sink, nlines);
    (void)hipDeviceSynchronize();
    window13<<<dim3((unsigned)((nwin * 192 + 191) / 192)), dim3(192)>>>(buf, sink, nwin);
    (void)hipDeviceSynchronize();
  }
  std::printf("buffer %zu bytes = %zu lines of 128 B; stream16 touches every byte; line1 one byte of each of the %zu lines; "
              "window13: %zu windows x 13 rows in %zu distinct lines (%zu bytes of pixels)\n",
              N, nlines, nlines, nwin, (N / (512 * 512)) * 31 * 13 * 4, nwin * 169);
  (void)hipFree(buf);
  (void)hipFree(sink);
  return 0;
}
