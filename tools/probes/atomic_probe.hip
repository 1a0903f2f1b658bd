// What do device-scope 64-bit integer atomics to a FEW addresses cost when every workgroup of a launch issues them at its end?
// (fixed-point BatchNorm statistics accumulated by the producing kernel's epilogue instead of partial rows + a finalize launch)
// Build: hipcc --offload-arch=gfx950 -O2 atomic_probe.hip -o atomic_probe ; run: ./atomic_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
// every workgroup spins `work` clocks (a stand-in for its tile), then `naddr` of its threads add one value each to acc[naddr]
template <int MODE>
__global__ void probe(unsigned long long* acc, float* partial, int naddr, int work) {
  long t0 = clock64();
  while (clock64() - t0 < work) {}
  const int t = threadIdx.x;
  if (t < naddr) {
    const long long v = (long long)((float)(blockIdx.x + t) * 16777216.f);
    if (MODE == 1) __hip_atomic_fetch_add(acc + t, (unsigned long long)v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (MODE == 2) partial[(long)blockIdx.x * naddr + t] = (float)v;
  }
}
__global__ void fin(const float* partial, float* out, int nblk, int naddr) {   // the existing second stage: one thread column per address
  const int c = blockIdx.x * 2 + (threadIdx.x >> 7);
  if (c >= naddr) return;
  float s = 0.f;
  for (int r = threadIdx.x & 127; r < nblk; r += 128) s += partial[(long)r * naddr + c];
  __shared__ float red[256];
  red[threadIdx.x] = s;
  __syncthreads();
  if ((threadIdx.x & 127) == 0) { float a = 0.f; for (int i = 0; i < 128; ++i) a += red[threadIdx.x + i]; out[c] = a; }
}
int main() {
  unsigned long long* acc; float* partial; float* out;
  hipMalloc(&acc, 4096 * 8); hipMalloc(&partial, 8192L * 4096 * 4); hipMalloc(&out, 4096 * 4);
  hipMemset(acc, 0, 4096 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int grids[] = {1568, 392, 6272}, addrs[] = {128, 256, 1024};
  for (int gi = 0; gi < 3; ++gi)
    for (int ai = 0; ai < 3; ++ai)
      for (int work = 0; work <= 20000; work += 20000) {
        float ms[4];
        for (int mode = 0; mode < 4; ++mode) {
          const int g = grids[gi], a = addrs[ai];
          for (int it = 0; it < 12; ++it) {
            if (it == 2) hipEventRecord(e0, 0);
            if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(g), dim3(256), 0, 0, acc, partial, a > 256 ? 256 : a, work);
            if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(g), dim3(256), 0, 0, acc, partial, a > 256 ? 256 : a, work);
            if (mode >= 2) hipLaunchKernelGGL(probe<2>, dim3(g), dim3(256), 0, 0, acc, partial, a > 256 ? 256 : a, work);
            if (mode == 3) hipLaunchKernelGGL(fin, dim3((a + 1) / 2), dim3(256), 0, 0, partial, out, g, a > 256 ? 256 : a);
          }
          hipEventRecord(e1, 0); hipEventSynchronize(e1);
          hipEventElapsedTime(&ms[mode], e0, e1);
          ms[mode] *= 100.f;   // us per launch (10 timed)
        }
        printf("grid %5d  addresses %4d  tile work %5d clk:  no stats %6.1f us | u64 atomics %6.1f | partial rows %6.1f | partial rows + finalize launch %6.1f\n",
               grids[gi], addrs[ai] > 256 ? 256 : addrs[ai], work, ms[0], ms[1], ms[2], ms[3]);
      }
  unsigned long long h[4];
  hipMemcpy(h, acc, 32, hipMemcpyDeviceToHost);
  printf("acc[0..1] = %llu %llu\n", h[0], h[1]);
  return 0;
}
