#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int *out)
{ int lane = threadIdx.x;
  int v = lane * 10;
  int a = __builtin_amdgcn_update_dpp(-1, v, 0x13C, 0xf, 0xf, false);   // wave_ror:1
  int b = __builtin_amdgcn_update_dpp(-1, v, 0x134, 0xf, 0xf, false);   // wave_rol:1
  int c = __builtin_amdgcn_update_dpp(-1, v, 0x138, 0xf, 0xf, false);   // wave_shr:1
  int d = __builtin_amdgcn_update_dpp(-1, v, 0x130, 0xf, 0xf, false);   // wave_shl:1
  out[lane * 4 + 0] = a; out[lane * 4 + 1] = b; out[lane * 4 + 2] = c; out[lane * 4 + 3] = d;
}
int main()
{ int *d; hipMalloc(&d, 64 * 4 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  int h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int l : {0, 1, 2, 31, 32, 62, 63}) printf("lane %2d: ror %4d rol %4d shr %4d shl %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
  return 0; }
