// Where do the workgroups of a persistent launch land?  (co-resident workgroups and the operand panels they could share.)
// 1024 workgroups of 256 threads with 36.8 KB of LDS each (bgemm_kernel's footprint: 4 per CU); every workgroup records HW_ID and XCC_ID.
// build: hipcc --offload-arch=gfx950 -O2 tools/probe/cu_map.hip -o gpurun_out/cu_map ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <map>

__global__ __launch_bounds__(256) void probe(unsigned* out, int spin) {
  extern __shared__ float lds[];
  lds[threadIdx.x] = (float)threadIdx.x;
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 4);      // HW_REG_HW_ID
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | (0 << 6) | 20);    // HW_REG_XCC_ID
    out[2 * blockIdx.x] = hw;
    out[2 * blockIdx.x + 1] = xcc;
  }
  // stay resident long enough for the whole grid to be placed (a persistent kernel's workgroups all are)
  float v = lds[(threadIdx.x + 1) & 255];
  for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;
  if (v == 123.456f) out[0] = 0;
}

int main() {
  const int grid = 1024;
  unsigned* d;
  hipMalloc(&d, grid * 2 * sizeof(unsigned));
  const size_t lds = 37 * 1024;
  hipFuncSetAttribute(reinterpret_cast<const void*>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(probe, dim3(grid), dim3(256), lds, 0, d, 200000);
    hipDeviceSynchronize();
  }
  std::vector<unsigned> h(grid * 2);
  hipMemcpy(h.data(), d, grid * 2 * sizeof(unsigned), hipMemcpyDeviceToHost);
  // gfx9 HW_ID: CU_ID [11:8], SH_ID [12], SE_ID [15:13]; XCC_ID [3:0]
  std::map<unsigned, std::vector<int> > by_cu;
  int xcc_ok = 0;
  for (int b = 0; b < grid; ++b) {
    const unsigned hw = h[2 * b], xcc = h[2 * b + 1] & 15;
    const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    if ((int)xcc == (b & 7)) ++xcc_ok;
    by_cu[(xcc << 8) | (se << 5) | (sh << 4) | cu].push_back(b);
  }
  printf("workgroups whose XCC_ID == blockIdx %% 8: %d of %d\n", xcc_ok, grid);
  printf("distinct (xcc, se, sh, cu): %zu\n", by_cu.size());
  int shown = 0;
  for (auto& kv : by_cu) {
    if ((kv.first >> 8) != 0) continue;                      // XCD 0
    printf("xcc %u se %u sh %u cu %2u : slots (blockIdx >> 3)", kv.first >> 8, (kv.first >> 5) & 7, (kv.first >> 4) & 1, kv.first & 15);
    for (int b : kv.second) printf(" %3d", b >> 3);
    printf("\n");
    if (++shown >= 40) break;
  }
  hipFree(d);
  return 0;
}
