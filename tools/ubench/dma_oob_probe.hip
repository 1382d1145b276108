// Does an out-of-range lane of `buffer_load_dwordx4 ... offen lds` (LDS-DMA) write ZEROS to its LDS slot?  (The conv gather of
// the planes GEMM wants padding pixels as zeros without a zero page.)   hipcc --offload-arch=gfx950 -O3 ... -o build/dma_oob_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define BLDS16(rsrc, lp, voff, soff) \
  __builtin_amdgcn_raw_ptr_buffer_load_lds((rsrc), (__attribute__((address_space(3))) void*)(lp), 16, (voff), (soff), 0, 0)
__global__ void probe(const uint32_t* src, uint32_t* out, int nrec, int soff) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[256];
  for (int i = threadIdx.x; i < 256; i += 64) lds[i] = 0xdeadbeefu;
  __syncthreads();
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t*>(src), 0, nrec, 0x00020000);
  const int lane = threadIdx.x;
  const int voff = (lane & 1) ? 0x7ffffff0 : lane * 16;      // odd lanes out of range
  BLDS16(r, lds, voff, soff);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 256; i += 64) out[i] = lds[i];
}
int main() {
  uint32_t *src, *out, h[256], hs[1024];
  hipMalloc(&src, 4096 * 4); hipMalloc(&out, 256 * 4);
  for (int i = 0; i < 1024; ++i) hs[i] = 0x1000 + i;
  hipMemcpy(src, hs, 4096, hipMemcpyHostToDevice);
  for (int cfg = 0; cfg < 2; ++cfg) {
    const int nrec = cfg == 0 ? 0x7fffffe0 : 2048, soff = cfg == 0 ? 0 : 1024;
    probe<<<1, 64>>>(src, out, nrec, soff);
    hipDeviceSynchronize();
    hipMemcpy(h, out, 1024, hipMemcpyDeviceToHost);
    int okv = 0, zer = 0, other = 0;
    for (int l = 0; l < 64; ++l)
      for (int j = 0; j < 4; ++j) {
        const uint32_t v = h[l * 4 + j];
        if (l & 1) { if (v == 0) ++zer; else ++other; }
        else { if (v == (uint32_t)(0x1000 + (soff / 4) + l * 4 + j)) ++okv; else ++other; }
      }
    printf("cfg %d (num_records %#x, soffset %d): in-range correct %d/128, out-of-range zero %d/128, other %d (first odd lane word %#x)\n",
           cfg, nrec, soff, okv, zer, other, h[4]);
  }
  return 0;
}
