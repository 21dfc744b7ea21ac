// Probe of ds_read_b64_tr_b16 on gfx950: LDS element i (u16) holds the value i; every lane passes its own byte address and
// prints the four 16-bit values it gets back.  hipcc --offload-arch=gfx950 -O2 tr_probe.hip -o tr_probe && ./tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void probe(int mode, unsigned long long* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const int l = threadIdx.x;
  unsigned addr;
  if (mode == 0) addr = l * 8;                                  // lane l -> elements 4l .. 4l+3
  else if (mode == 1) addr = (l & 15) * 2 * 1 + (l >> 4) * 128; // the guide's formula: element (l&15) + (l>>4)*64  (byte address, not 8-aligned!)
  else if (mode == 2) addr = (l & 15) * 64 + (l >> 4) * 8;      // row-major [16 rows][32 elems]: lane -> row (l&15), 4-element column block (l>>4)
  else addr = (l & 3) * 64 + ((l >> 2) & 3) * 8 + (l >> 4) * 1024;   // 4 lanes = 4 rows of 32 elems, 4 column blocks, 4 groups
  addr += (unsigned)(size_t)(__attribute__((address_space(3))) void*)lds;
  unsigned long long v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  out[l] = v;
}
int main() {
  unsigned long long* d; hipMalloc(&d, 64 * 8);
  unsigned long long h[64];
  for (int mode = 0; mode < 4; ++mode) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, mode, d);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) printf("  lane %2d: %4u %4u %4u %4u\n", l, (unsigned)(h[l] & 0xffff), (unsigned)((h[l] >> 16) & 0xffff), (unsigned)((h[l] >> 32) & 0xffff), (unsigned)(h[l] >> 48));
  }
  return 0;
}
