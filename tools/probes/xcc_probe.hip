// Which XCD does workgroup (x, z) of a (gx, 1, gz) grid run on?  Prints the XCC_ID hardware register per workgroup.
// Build: hipcc --offload-arch=gfx950 -O2 xcc_probe.hip -o xcc_probe ; run: ./xcc_probe gx gz
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ void probe(int* out) {
  if (threadIdx.x == 0) {
    const int xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15;   // HW_REG_XCC_ID[3:0]
    out[blockIdx.z * gridDim.x + blockIdx.x] = xcc;
  }
  // keep the workgroup resident for a moment so that later workgroups cannot reuse its slot out of order
  long t0 = clock64();
  while (clock64() - t0 < 20000) {}
}
int main(int argc, char** argv) {
  const int gx = argc > 1 ? atoi(argv[1]) : 5, gz = argc > 2 ? atoi(argv[2]) : 16;
  int* d; hipMalloc(&d, gx * gz * sizeof(int));
  hipLaunchKernelGGL(probe, dim3(gx, 1, gz), dim3(256), 0, 0, d);
  int* h = (int*)malloc(gx * gz * sizeof(int));
  hipMemcpy(h, d, gx * gz * sizeof(int), hipMemcpyDeviceToHost);
  int agree = 0;
  for (int z = 0; z < gz; ++z) {
    if (z < 8) { printf("z=%d:", z); for (int x = 0; x < gx; ++x) printf(" %d", h[z * gx + x]); printf("\n"); }
    for (int x = 0; x < gx; ++x) agree += (h[z * gx + x] == ((z * gx + x) & 7));
  }
  printf("linear-id %% 8 == XCC_ID for %d of %d workgroups\n", agree, gx * gz);
  return 0;
}
