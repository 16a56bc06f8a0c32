#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void k(uint32_t * out)
{
  const uint32_t lane = threadIdx.x;
  const uint32_t v = 100u + lane;
  out[lane] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x138, 0xF, 0xF, true);      // wave_shr:1
  out[64 + lane] = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x130, 0xF, 0xF, true); // wave_shl:1
  out[128 + lane] = (uint32_t)__builtin_amdgcn_update_dpp(7777, (int)v, 0x138, 0xF, 0xF, false);
  out[192 + lane] = (uint32_t)__builtin_amdgcn_update_dpp(7777, (int)v, 0x130, 0xF, 0xF, false);
}
int main()
{
  uint32_t * d; hipMalloc(&d, 256 * 4);
  k<<<1, 64>>>(d);
  uint32_t h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  for (int a = 0; a < 4; a++) { for (int i = 0; i < 64; i++) printf("%u ", h[a * 64 + i]); printf("\n"); }
  return 0;
}
