// membench.hip -- what the memory system gives for the access patterns the organised-scan kernel could use.
// Input: `scans` scans of R x C 32-byte records, column-major (record (c, r) of a scan at ((c * R) + r) * 32).
// Every pattern reads each record's first 16 bytes and the dword at +20, exactly once, and writes only a checksum.
//   build: hipcc --offload-arch=gfx950 -O3 -o membench membench.hip ;  run: ./membench [scans]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

constexpr int R = 64, C = 1800, B = 6, SPAN = 320;   // six blocks of ~300 columns, 320 positions loaded per unit (halo included)

__device__ inline void consume(float4 v, uint32_t w, float & acc) {acc += v.x + v.y + v.z + (float)(w & 0xFFFFu);}

// G rings per workgroup (G = 4, 8, 16, 64): lane = (column piece, ring of the group); 256 threads.
// XCD: 0 = consecutive workgroups take adjacent ring groups (they land on different XCDs: blocks are dealt round-robin
// over the 8 XCDs); 1 = the workgroups that share an XCD (ids b, b + 8, ...) take adjacent ring groups
template<int G, int XCD = 0>
__global__ __launch_bounds__(256) void read_groups(const uint8_t * __restrict__ pts, float * __restrict__ out)
{
  constexpr int groups = R / G;
  constexpr int cols_per_instr = 256 / G;                   // columns one pass of the workgroup covers
  constexpr int span = SPAN * 4 / G;                        // columns per workgroup so that bytes per workgroup stay equal
  constexpr int units = (C + span - 1) / span;              // workgroups along the columns
  const uint32_t s = blockIdx.y;
  uint32_t bx = blockIdx.x;
  if (XCD) {
    // ids of one XCD: x = 8 q + r (r fixed).  Per XCD the sequence q = 0, 1, 2 ... should walk g fastest.
    const uint32_t nb = gridDim.x, r = bx % 8u, q = bx / 8u, per = nb / 8u;      // nb is a multiple of 8 here
    bx = r * per + q;                                                           // XCD r owns the contiguous range [r * per, (r + 1) * per)
  }
  const uint32_t g = bx % groups, j = bx / groups;
  if (j >= units) {return;}
  const uint32_t t = threadIdx.x, sub = t % G, cq = t / G;
  const uint8_t * base = pts + (size_t)s * R * C * 32;
  float acc = 0.f;
  float4 v[span / cols_per_instr];
  uint32_t w[span / cols_per_instr];
#pragma unroll
  for (int m = 0; m < span / cols_per_instr; m++) {
    uint32_t c = j * span + m * cols_per_instr + cq;
    c = c < (uint32_t)C ? c : C - 1;
    const uint8_t * p = base + ((size_t)c * R + g * G + sub) * 32;
    v[m] = *reinterpret_cast<const float4 *>(p);
    w[m] = *reinterpret_cast<const uint32_t *>(p + 20);
  }
#pragma unroll
  for (int m = 0; m < span / cols_per_instr; m++) {consume(v[m], w[m], acc);}
  if (acc == 1.2345e30f) {out[0] = acc;}
}

// linear: consecutive threads read consecutive records
__global__ __launch_bounds__(256) void read_linear(const uint8_t * __restrict__ pts, float * __restrict__ out, size_t n)
{
  float acc = 0.f;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += 5 * stride) {
    float4 v[5]; uint32_t w[5];
#pragma unroll
    for (int m = 0; m < 5; m++) {
      size_t i = i0 + m * stride; i = i < n ? i : n - 1;
      v[m] = *reinterpret_cast<const float4 *>(pts + i * 32);
      w[m] = *reinterpret_cast<const uint32_t *>(pts + i * 32 + 20);
    }
#pragma unroll
    for (int m = 0; m < 5; m++) {consume(v[m], w[m], acc);}
  }
  if (acc == 1.2345e30f) {out[0] = acc;}
}

// the outputs of the unit kernel: per point 1 byte + 8 bytes, ring-major; one wave per (ring, block) unit of 300 positions
__global__ __launch_bounds__(256) void write_outputs(uint8_t * __restrict__ lab, double * __restrict__ cur, int with_read, const uint8_t * __restrict__ pts)
{
  const uint32_t s = blockIdx.y, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const uint32_t groups = R / 4, g = blockIdx.x % groups, j = blockIdx.x / groups;
  const uint32_t ring = 4 * g + wave;
  const size_t off = ((size_t)s * R + ring) * 1856 + j * 300;
  for (int k = 0; k < 5; k++) {
    const uint32_t q = 64 * k + lane;
    if (q < 300) {
      lab[off + q] = (uint8_t)(q & 7);
      cur[off + q] = (double)q;
    }
  }
}

int main(int argc, char ** argv)
{
  const int scans = argc > 1 ? atoi(argv[1]) : 1024;
  const size_t n = (size_t)scans * R * C, bytes = n * 32;
  uint8_t * pts; float * out; uint8_t * lab; double * cur;
  hipMalloc(&pts, bytes); hipMalloc(&out, 64);
  hipMalloc(&lab, (size_t)scans * R * 1856); hipMalloc(&cur, (size_t)scans * R * 1856 * 8);
  hipMemset(pts, 1, bytes);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  auto time = [&](const char * name, auto launch, double gb) {
    launch(); hipDeviceSynchronize();
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
      hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b); best = ms < best ? ms : best;
    }
    printf("%-34s %8.1f us  %7.1f GB/s\n", name, best * 1e3, gb / (best * 1e-3));
  };
  const double gb = bytes / 1e9;
  time("linear (grid 8192)", [&] {hipLaunchKernelGGL(read_linear, dim3(8192), dim3(256), 0, 0, pts, out, n);}, gb);
  time("4 rings per workgroup", [&] {hipLaunchKernelGGL(read_groups<4>, dim3(16 * 6, scans), dim3(256), 0, 0, pts, out);}, gb * 320 * 6 / 1800);
  time("4 rings per workgroup, XCD-local", [&] {hipLaunchKernelGGL((read_groups<4, 1>), dim3(16 * 6, scans), dim3(256), 0, 0, pts, out);}, gb * 320 * 6 / 1800);
  time("8 rings per workgroup", [&] {hipLaunchKernelGGL(read_groups<8>, dim3(8 * 12, scans), dim3(256), 0, 0, pts, out);}, gb * 160 * 12 / 1800);
  time("16 rings per workgroup", [&] {hipLaunchKernelGGL(read_groups<16>, dim3(4 * 23, scans), dim3(256), 0, 0, pts, out);}, gb * 80 * 23 / 1800);
  time("64 rings per workgroup", [&] {hipLaunchKernelGGL(read_groups<64>, dim3(1 * 90, scans), dim3(256), 0, 0, pts, out);}, gb * 20 * 90 / 1800);
  const double wgb = (double)scans * R * 1800 * 9 / 1e9;
  time("outputs: 1 B + 8 B per point", [&] {hipLaunchKernelGGL(write_outputs, dim3(16 * 6, scans), dim3(256), 0, 0, lab, cur, 0, pts);}, wgb);
  return 0;
}
