// HBM stream ceilings on this box, for pricing the pointwise convs: float4 copy, read-only, write-only and the
// read:write mixes of the expanding / contracting 1x1 convs (16->96 = 1:6, 96->16 = 6:1, 32->64 = 1:2).
// usage: tools/build/membench   (prints GB/s per pattern; buffers of 1 GiB, beyond the 256 MiB Infinity Cache)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1);} } while (0)

// each workgroup walks chunks of (R + W) * 4 KB: R float4 loads, W float4 stores per thread
template <int R, int W>
__global__ __launch_bounds__(256) void mix_kernel(const float4* __restrict__ src, float4* __restrict__ dst, size_t n_units) {
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  for (size_t u = blockIdx.x; u < n_units; u += gridDim.x) {
    float4 v[R > 0 ? R : 1];
#pragma unroll
    for (int i = 0; i < R; ++i) v[i] = src[(u * R + i) * 256 + threadIdx.x];
#pragma unroll
    for (int i = 0; i < R; ++i) { acc.x += v[i].x; acc.y += v[i].y; acc.z += v[i].z; acc.w += v[i].w; }
#pragma unroll
    for (int i = 0; i < W; ++i) dst[(u * W + i) * 256 + threadIdx.x] = make_float4(acc.x + i, acc.y, acc.z, acc.w);
  }
  if (W == 0 && acc.x == 12345.678f) dst[0] = acc;
}

// The store patterns of an MFMA epilogue over rows of ROWB bytes (one pixel = ROWB bytes, pixels contiguous):
// SEG = 64: an instruction writes 16 pixels x 64 B (lane = 16 pixel lanes x 4 chunks - a 16-channel tile);
// SEG = 128: 8 pixels x 128 B (two adjacent tiles after the row_ror:8 exchange).  Every wave writes whole rows
// in the end (all segments of its 16 pixels, one after the other), as a conv epilogue does.
template <int SEG, int ROWB>
__global__ __launch_bounds__(256) void seg_write_kernel(char* __restrict__ dst, size_t n_px16) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (size_t t = (size_t)blockIdx.x * 4 + wave; t < n_px16; t += (size_t)gridDim.x * 4) {
    char* base = dst + t * 16 * ROWB;
    if (SEG == 64) {
#pragma unroll
      for (int sgm = 0; sgm < ROWB / 64; ++sgm)
        *reinterpret_cast<float4*>(base + (size_t)(lane & 15) * ROWB + sgm * 64 + (lane >> 4) * 16) =
            make_float4(1.f, 2.f, 3.f, (float)sgm);
    } else {
#pragma unroll
      for (int sgm = 0; sgm < ROWB / 128; ++sgm) {
#pragma unroll
        for (int half = 0; half < 2; ++half)
          *reinterpret_cast<float4*>(base + (size_t)((lane & 7) + 8 * half) * ROWB + sgm * 128 + (lane >> 3) * 16) =
              make_float4(1.f, 2.f, 3.f, (float)sgm);
      }
    }
  }
}
template <int SEG, int ROWB>
void run_seg(const char* name, char* dst, size_t bytes, int grid) {
  const size_t n = bytes / (16 * ROWB);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((seg_write_kernel<SEG, ROWB>), dim3(grid), dim3(256), 0, 0, dst, n);
  CK(hipEventRecord(e0, 0));
  const int reps = 5;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((seg_write_kernel<SEG, ROWB>), dim3(grid), dim3(256), 0, 0, dst, n);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double moved = (double)n * 16 * ROWB * reps;
  printf("%-28s grid %5d: %8.1f GB/s  (%.1f us per launch)\n", name, grid, moved / (ms * 1e-3) / 1e9, ms * 1e3 / reps);
}

template <int R, int W>
void run(const char* name, const float4* src, float4* dst, size_t bytes, int grid) {
  const size_t per_unit = (size_t)(R > W ? R : W) * 4096;  // the larger side bounds the unit count
  const size_t n_units = bytes / per_unit;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((mix_kernel<R, W>), dim3(grid), dim3(256), 0, 0, src, dst, n_units);
  CK(hipEventRecord(e0, 0));
  const int reps = 5;
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((mix_kernel<R, W>), dim3(grid), dim3(256), 0, 0, src, dst, n_units);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double moved = (double)n_units * (R + W) * 4096.0 * reps;
  printf("%-28s grid %5d: %8.1f GB/s  (%.1f us per launch, %.0f MB)\n", name, grid, moved / (ms * 1e-3) / 1e9, ms * 1e3 / reps,
         moved / reps / 1e6);
}

int main() {
  const size_t bytes = (size_t)1 << 30;
  float4 *a, *b;
  CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
  CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes));
  for (int grid : {512, 1024, 2048, 4096}) {
    run<4, 4>("copy 1:1 (4+4 per thread)", a, b, bytes, grid);
    run<8, 0>("read only (8)", a, b, bytes, grid);
    run<0, 8>("write only (8)", a, b, bytes, grid);
    run<1, 6>("expand 1:6 (16->96)", a, b, bytes, grid);
    run<6, 1>("contract 6:1 (96->16)", a, b, bytes, grid);
    run<2, 4>("1:2 (32->64)", a, b, bytes, grid);
    run<4, 2>("2:1 (64->32)", a, b, bytes, grid);
    run_seg<64, 384>("write 64 B segs, 384 B rows", (char*)b, bytes, grid);
    run_seg<128, 384>("write 128 B segs, 384 B rows", (char*)b, bytes, grid);
    run_seg<64, 256>("write 64 B segs, 256 B rows", (char*)b, bytes, grid);
    run_seg<128, 256>("write 128 B segs, 256 B rows", (char*)b, bytes, grid);
  }
  return 0;
}
