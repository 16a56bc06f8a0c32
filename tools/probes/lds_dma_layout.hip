#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>
// layout probe: where does lane l's data land for global_load_lds_dwordx3 / x4 / dword?
template<int BYTES>
__global__ void probe(const uint8_t * src, uint32_t * out)
{
  __shared__ __attribute__((aligned(16))) uint32_t zone[1024];
  for (int i = threadIdx.x; i < 1024; i += 64) zone[i] = 0xDEADBEEFu;
  __syncthreads();
  const uint8_t * p = src + threadIdx.x * 32u;
  const uint32_t dst = (uint32_t)reinterpret_cast<uintptr_t>(&zone[0]);
  uint32_t keep;
  if (BYTES == 12)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx3 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(p), "s"(dst) : "memory");
  else if (BYTES == 16)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(p), "s"(dst) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off offset:20\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(p), "s"(dst) : "memory");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 1024; i += 64) out[i] = zone[i];
}
int main()
{
  std::vector<uint32_t> h(64 * 8);
  for (int l = 0; l < 64; l++) for (int d = 0; d < 8; d++) h[l * 8 + d] = (l << 8) | d;
  uint8_t * src; uint32_t * out;
  hipMalloc(&src, h.size() * 4); hipMalloc(&out, 4096);
  hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  std::vector<uint32_t> r(1024);
  for (int v = 0; v < 3; v++) {
    if (v == 0) hipLaunchKernelGGL(probe<12>, dim3(1), dim3(64), 0, 0, src, out);
    if (v == 1) hipLaunchKernelGGL(probe<16>, dim3(1), dim3(64), 0, 0, src, out);
    if (v == 2) hipLaunchKernelGGL(probe<4>, dim3(1), dim3(64), 0, 0, src, out);
    hipDeviceSynchronize();
    hipMemcpy(r.data(), out, 4096, hipMemcpyDeviceToHost);
    printf("variant %d:", v);
    for (int i = 0; i < 24; i++) printf(" %x", r[i]);
    int last = 0; for (int i = 0; i < 1024; i++) if (r[i] != 0xDEADBEEFu) last = i;
    printf(" ... last written dword %d\n", last);
    // lane stride: find where lane 1's first dword (0x100) is
    for (int i = 0; i < 1024; i++) if (r[i] == 0x100u || r[i] == 0x105u) {printf("  lane1 first dword at %d\n", i); break;}
    for (int i = 0; i < 1024; i++) if (r[i] == 0x3F00u || r[i] == 0x3F05u) {printf("  lane63 first dword at %d\n", i); break;}
  }
  return 0;
}
