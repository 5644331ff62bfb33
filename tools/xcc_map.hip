// Scratch: which XCD does workgroup i of a 256 x 1024-thread launch run on?  (HW_REG_XCC_ID, gfx940+)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(1024) void k(int* out) {
  if (threadIdx.x == 0) out[blockIdx.x] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15;
}
int main() {
  int* d; hipMalloc(&d, 4096 * 4);
  for (int grid : {256, 252, 192}) {
    hipMemset(d, 0xff, 4096 * 4);
    k<<<grid, 1024, 65536>>>(d);
    hipDeviceSynchronize();
    int h[4096]; hipMemcpy(h, d, grid * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < grid; i++) if (h[i] != h[i & 7]) bad++;
    printf("grid %d: xcc of wg 0..15:", grid);
    for (int i = 0; i < 16; i++) printf(" %d", h[i]);
    printf("  | workgroups not on the XCD of (i %% 8): %d\n", bad);
  }
  return 0;
}
