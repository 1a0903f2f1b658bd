// Hardware-semantics probe for gfx950 (run on the GPU box): LDS-DMA zero fill of out-of-range lanes and the
// ds_read_b64_tr_b16 lane mapping.  Prints what the hardware does so kernels are written against facts.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) short s4;
#define LDSP(p) ((__attribute__((address_space(3))) void*)(p))

__global__ void probe_dma(const uint32_t* g, uint32_t* out, int nbytes) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[512];
  for (int i = threadIdx.x; i < 512; i += 64) lds[i] = 0xABABABABu;
  __syncthreads();
  auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)g, 0, nbytes, 0x00020000);
  const int lane = threadIdx.x;
  const uint32_t voff = (lane & 1) ? 0x7ffffff0u : (uint32_t)(lane * 16);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDSP(lds), 16, voff, 0, 0, 0);
  // second DMA into the upper KB with an immediate offset on the source
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDSP(lds + 256), 16, (uint32_t)(lane * 16), 0, 16, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 64) out[i] = lds[i];
}

__global__ void probe_tr(uint16_t* out) {
  __shared__ __attribute__((aligned(16))) uint16_t lds[64 * 64];
  for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (uint16_t)i;   // value = row*64 + col
  __syncthreads();
  const int lane = threadIdx.x, g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  const int row = 4 * g + q, col = 4 * p;
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(lds + row * 64 + col));
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (uint16_t)v[j];
}

int main() {
  uint32_t *g, *o; uint16_t* o2;
  hipMalloc(&g, 4096); hipMalloc(&o, 2048); hipMalloc(&o2, 64 * 4 * 2);
  uint32_t h[1024]; for (int i = 0; i < 1024; ++i) h[i] = 0x1000 + i;
  hipMemcpy(g, h, 4096, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe_dma, dim3(1), dim3(64), 0, 0, g, o, 1024);
  uint32_t r[512]; hipMemcpy(r, o, 2048, hipMemcpyDeviceToHost);
  printf("DMA lane chunks (first dword of each 16B chunk; even lanes in range, odd lanes out of range):\n");
  for (int l = 0; l < 8; ++l) printf(" lane%d:%08x", l, r[l * 4]);
  printf("\n second DMA (imm offset 16): lane0:%08x lane1:%08x lane62:%08x lane63:%08x (lane63 source is past num_records)\n", r[256], r[260], r[256 + 62 * 4], r[256 + 63 * 4]);
  hipLaunchKernelGGL(probe_tr, dim3(1), dim3(64), 0, 0, o2);
  uint16_t t[256]; hipMemcpy(t, o2, 512, hipMemcpyDeviceToHost);
  printf("tr16_b64: lane -> 4 x (row,col) received; lane 4q+p of group g supplied address of row 4g+q, cols 4p..4p+3\n");
  for (int l = 0; l < 64; ++l) {
    printf(" L%02d:", l);
    for (int j = 0; j < 4; ++j) printf("(%d,%d)", t[l * 4 + j] / 64, t[l * 4 + j] % 64);
    if (l % 4 == 3) printf("\n");
  }
  return 0;
}
