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

// The unit kernel's memory operations with nothing between them but a wait: workgroup = block j of G adjacent rings (grid
// = groups x blocks x scans as ring_unit_org_kernel's), one ring per wave (64 G threads), every lane asks for 5 x (16 + 4)
// bytes, the workgroup meets at a barrier, each wave then "computes" for `work` rounds of dependent vector instructions and
// stores its ring's 300 labels and curvatures and ~12 % of them as 20-byte feature records.  MAP: 0 = ring group =
// blockIdx.x; 1 = turned by the scan index; 2 = two adjacent groups per XCD, turned.  Dynamic LDS sets the workgroups per CU.
// OUT: bit 0 labels, bit 1 curvature, bit 2 records.
template<int MAP, int G = 4, int OUT = 7>
__global__ __launch_bounds__(64 * G) void unit_memory(const uint8_t * __restrict__ pts, uint8_t * __restrict__ lab, double * __restrict__ cur,
  float4 * __restrict__ rec, uint32_t * __restrict__ idx, float * __restrict__ out, int work)
{
  extern __shared__ uint8_t dyn[];
  constexpr uint32_t groups = R / G;
  const uint32_t s = blockIdx.z, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  uint32_t j = blockIdx.y;
  uint32_t g = blockIdx.x;
  if (MAP == 1) {g = (g + s) % groups;}
  if (MAP == 2) {g = (groups >= 16 ? ((((g & 7u) << 1) | (g >> 3)) + 2u * s) : (g + s)) % groups;}
  if (MAP == 3 || MAP == 4) {
    // an XCD (x mod 8) takes TWELVE consecutive (group, block) pairs of the scan, group fastest -- up to 1.5 KB of every
    // column -- and which twelve turns with the scan (3), or stays (4)
    const uint32_t r = blockIdx.x % 8u, q = blockIdx.x / 8u + (gridDim.x / 8u) * blockIdx.y, per = gridDim.x * gridDim.y / 8u;
    const uint32_t b = ((MAP == 3 ? r + s : r) % 8u) * per + q;
    g = b % groups;
    j = b / groups;
  }
  const uint32_t sub = lane % G, cq = lane / G;
  constexpr uint32_t per_wave = 64 / G;              // columns one instruction of a wave covers
  const uint8_t * base = pts + (size_t)s * R * C * 32;
  float4 v[5]; uint32_t w[5];
  // OUT & 8192, the records' second leg inside the kernel: the unit's slot of the scan kDuty scans earlier is read with
  // the unit's own records (its latency under the same wait) and written to the dense clouds at the end
  constexpr uint32_t kDuty = 32, kDenseAt = 64u << 20;
  float4 duty_p = make_float4(0.f, 0.f, 0.f, 0.f); uint32_t duty_i = 0;
  const bool duty = (OUT & 8192) && s >= kDuty && lane < 38;
  if (duty) {
    const float4 * sl = rec + ((((size_t)(s - kDuty) * R + G * g + wave) * 6 + j) * 80);
    duty_p = sl[lane];
    duty_i = reinterpret_cast<const uint32_t *>(sl + 64)[lane];
  }
#pragma unroll
  for (int m = 0; m < 5; m++) {
    uint32_t c = j * 298 + 64 * m + per_wave * wave + cq;
    c = c < (uint32_t)C ? c : C - 1;
    const uint8_t * p = base + ((size_t)c * R + G * g + sub) * 32;
    v[m] = *reinterpret_cast<const float4 *>(p);
    w[m] = *reinterpret_cast<const uint32_t *>(p + 20);
  }
  float acc = 0.f;
#pragma unroll
  for (int m = 0; m < 5; m++) {consume(v[m], w[m], acc);}
  reinterpret_cast<float *>(dyn)[threadIdx.x] = acc;
  __syncthreads();
  acc += reinterpret_cast<float *>(dyn)[threadIdx.x ^ 64];
  for (int i = 0; i < work; i++) {
#pragma unroll
    for (int u = 0; u < 16; u++) {asm volatile ("v_fma_f32 %0, %0, %0, %0" : "+v"(acc));}
  }
  const uint32_t ring = G * g + wave;
  const size_t off = ((size_t)s * R + ring) * 1856 + j * 300;
  const bool feat = (lane & 7u) == 3u;             // one position in eight: ~12 % feature points
#pragma unroll
  for (int k = 0; k < 5; k++) {
    const uint32_t q = 64 * k + lane;
    if (q < 300) {
      if (OUT & 1) {lab[off + q] = (uint8_t)(q & 7);}
      if (OUT & 2) {cur[off + q] = (double)acc;}
      if ((OUT & 4) && feat) {
        rec[off + 8 * k + (lane >> 3)] = make_float4(acc, acc, acc, acc);
        idx[off + 8 * k + (lane >> 3)] = q;
      }
      if ((OUT & 2048) && feat) {       // 32-byte records chunk by chunk into the unit's slot of 64
        float4 * r32 = rec + 2 * ((((size_t)s * R + ring) * 6 + j) * 64 + 8 * k + (lane >> 3));
        r32[0] = make_float4(acc, acc, acc, acc);
        r32[1] = make_float4(__uint_as_float(q), 0.f, 0.f, 0.f);
      }
      if ((OUT & 8) && feat) {          // one 32-byte record per feature (two float4 halves of one sector, one instruction pair)
        float4 * r32 = rec + 2 * (off + 8 * k + (lane >> 3));
        r32[0] = make_float4(acc, acc, acc, acc);
        r32[1] = make_float4(__uint_as_float(q), 0.f, 0.f, 0.f);
      }
    }
  }
  if (OUT & 16) {                       // all of the unit's ~38 records by ONE instruction per array at the end
    if (lane < 38) {
      rec[off + lane] = make_float4(acc, acc, acc, acc);
      idx[off + lane] = lane;
    }
  }
  if (OUT & 4096) {                     // as built: a plane of 16-byte points and a plane of indices in the unit's slot of 1 280 bytes
    float4 * sl = rec + ((((size_t)s * R + ring) * 6 + j) * 80);
    if (lane < 38) {
      sl[lane] = make_float4(acc, acc, acc, acc);
      reinterpret_cast<uint32_t *>(sl + 64)[lane] = lane;
    }
  }
  if (OUT & 32768) {                    // ... the plane of indices right behind the points that are there (one run of 20 n bytes)
    float4 * sl = rec + ((((size_t)s * R + ring) * 6 + j) * 80);
    if (lane < 38) {
      sl[lane] = make_float4(acc, acc, acc, acc);
      reinterpret_cast<uint32_t *>(sl + 38)[lane] = lane;
    }
  }
  if (OUT & 65536) {                    // ... slots of 48 records (960 bytes), indices right behind the points
    float4 * sl = rec + ((((size_t)s * R + ring) * 6 + j) * 60);
    if (lane < 38) {
      sl[lane] = make_float4(acc, acc, acc, acc);
      reinterpret_cast<uint32_t *>(sl + 38)[lane] = lane;
    }
  }
  if (duty) {
    const size_t at = (((size_t)(s - kDuty) * R + ring) * 6 + j) * 38 + lane;
    rec[kDenseAt + at] = duty_p;
    idx[at] = duty_i;
  }
  if (OUT & 16384) {                    // the records straight into dense clouds (what a unit that knew its place would write)
    const size_t at = (((size_t)s * R + ring) * 6 + j) * 38 + lane;
    if (lane < 38) {
      rec[kDenseAt + at] = make_float4(acc, acc, acc, acc);
      idx[at] = lane;
    }
  }
  if (OUT & (64 | 128)) {               // records where a bump allocator puts them (64: one counter per scan, 128: one in all): dense in time
    uint32_t * counter = reinterpret_cast<uint32_t *>(out) + 16 + ((OUT & 64) ? s : 0u);
    uint32_t at = 0;
    if (lane == 0) {at = atomicAdd(counter, 38u);}
    at = __builtin_amdgcn_readfirstlane(at) % (R * 1856u - 64u);      // (38 records from `at` stay inside the scan's part of the arrays)
    const size_t o2 = (size_t)s * R * 1856 + at;
    if (lane < 38) {
      rec[o2 + lane] = make_float4(acc, acc, acc, acc);
      idx[o2 + lane] = lane;
    }
  }
  if (OUT & 256) {                      // 32-byte records by one store, in a slot of 64 per unit (2 KB apart instead of 4.8)
    const size_t u2 = ((((size_t)s * R + ring) * 6 + j) * 64) * 2;
    if (lane < 76) {rec[u2 + lane] = make_float4(acc, acc, acc, acc);}
  }
  if (OUT & 512) {                      // 20-byte records (five dwords) packed, 38 of them = 190 dwords by three dword stores
    float * r20 = reinterpret_cast<float *>(rec) + (off * 5);
#pragma unroll
    for (int t = 0; t < 3; t++) {
      if (64 * t + lane < 190) {r20[64 * t + lane] = acc;}
    }
  }
  if (OUT & 1024) {                     // 32-byte records by one non-temporal store
    if (lane < 76) {__builtin_nontemporal_store(acc, reinterpret_cast<float *>(rec + 2 * off + lane)); 
      __builtin_nontemporal_store(acc, reinterpret_cast<float *>(rec + 2 * off + lane) + 1);
      __builtin_nontemporal_store(acc, reinterpret_cast<float *>(rec + 2 * off + lane) + 2);
      __builtin_nontemporal_store(acc, reinterpret_cast<float *>(rec + 2 * off + lane) + 3);}
  }
  if (OUT & 32) {                       // ... as 32-byte records: lanes 2 i and 2 i + 1 write the two halves of record i
    if (lane < 76) {
      rec[2 * off + lane] = make_float4(acc, acc, acc, acc);
    }
  }
  if (acc == 1.2345e30f) {out[0] = acc;}
}

// plain copies: U float4 in flight per thread, grid-stride
template<int U>
__global__ __launch_bounds__(256) void copy_f4(const float4 * __restrict__ src, float4 * __restrict__ dst, size_t n)
{
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += U * stride) {
    float4 v[U];
#pragma unroll
    for (int k = 0; k < U; k++) {const size_t j = i + k * stride; v[k] = src[j < n ? j : n - 1];}
#pragma unroll
    for (int k = 0; k < U; k++) {const size_t j = i + k * stride; if (j < n) {dst[j] = v[k];}}
  }
}

// The bucketing kernel's memory side: a chunk of 2 048 column-major points read once (16 + 4 bytes of each 32-byte record),
// written ring-major -- per chunk and ring a run of 32 positions.  AOS = 0: three staging arrays (x, y | z | index: runs of
// 256, 128 and 128 bytes, as ring_scatter_kernel writes them); 1: one array of 16-byte records {x, y, z, index} (runs of 512).
template<int AOS>
__global__ __launch_bounds__(256) void scatter_model(const uint8_t * __restrict__ pts, float2 * __restrict__ sxy, float * __restrict__ sz,
  uint32_t * __restrict__ sidx, float4 * __restrict__ sp)
{
  const uint32_t s = blockIdx.y, c = blockIdx.x, t = threadIdx.x;
  const uint8_t * base = pts + ((size_t)s * R * C + (size_t)c * 2048) * 32;
  float4 v[8]; uint32_t w[8];
#pragma unroll
  for (int k = 0; k < 8; k++) {
    // pass k: the chunk's points whose ring is 8 k + t / 32, position t % 32 of the run: record (32 c + t % 32) * 64 + ring
    const uint32_t ring = 8 * k + (t >> 5), pos = t & 31u;
    const uint32_t col = 32 * c + pos;
    const uint8_t * p = base + ((size_t)pos * R + ring) * 32;
    if (col < (uint32_t)C) {v[k] = *reinterpret_cast<const float4 *>(p); w[k] = *reinterpret_cast<const uint32_t *>(p + 20);} else {v[k] = make_float4(0, 0, 0, 0); w[k] = 0;}
  }
#pragma unroll
  for (int k = 0; k < 8; k++) {
    const uint32_t ring = 8 * k + (t >> 5), pos = t & 31u, col = 32 * c + pos;
    if (col < (uint32_t)C) {
      const size_t at = ((size_t)s * R + ring) * 1856 + col;
      if (AOS) {
        sp[at] = make_float4(v[k].x, v[k].y, v[k].z, __uint_as_float(w[k]));
      } else {
        sxy[at] = make_float2(v[k].x, v[k].y); sz[at] = v[k].z; sidx[at] = w[k];
      }
    }
  }
}

// The compaction's memory side: a wave per ring copies the ~33 used records of each of its six units' slots (1 280 bytes per
// slot, slots back to back; the points, then the indices -- BEHIND = 64: behind all 64 places of the points, as first built;
// 33: right behind the points that are there) into dense arrays (points 16 bytes, indices 4), no table lookups.
template<int BEHIND>
__global__ __launch_bounds__(256) void compact_model(const float4 * __restrict__ slots, float4 * __restrict__ pts_out, uint32_t * __restrict__ idx_out)
{
  const uint32_t s = blockIdx.y, ring = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  const size_t unit0 = ((size_t)s * R + ring) * 6;
  const size_t out0 = ((size_t)s * R + ring) * 200;            // 6 x 33 = 198 records per ring, dense
  float4 v[6]; uint32_t w[6];
#pragma unroll
  for (int u = 0; u < 6; u++) {
    const float4 * slot = slots + (unit0 + u) * 80;           // 1 280 bytes = 80 float4
    v[u] = slot[lane < 33 ? lane : 0];
    w[u] = reinterpret_cast<const uint32_t *>(slot + BEHIND)[lane < 33 ? lane : 0];
  }
#pragma unroll
  for (int u = 0; u < 6; u++) {
    if (lane < 33) {pts_out[out0 + 33 * u + lane] = v[u]; idx_out[out0 + 33 * u + lane] = w[u];}
  }
}

int main(int argc, char ** argv)
{
  const int scans = argc > 1 ? atoi(argv[1]) : 1024;
  const size_t n = (size_t)scans * R * C, bytes = n * 32;
  uint8_t * pts; float * out; uint8_t * lab; double * cur;
  hipMalloc(&pts, bytes); hipMalloc(&out, 64 + 4 * 4096); hipMemset(out, 0, 64 + 4 * 4096);
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
  {
    // a copy of 1 GiB inside the input buffer (read + written bytes counted)
    const size_t cb = (size_t)1 << 30, cn = cb / 16;
    const float4 * s4 = reinterpret_cast<const float4 *>(pts); float4 * d4 = reinterpret_cast<float4 *>(pts + cb);
    const int grids[] = {1024, 2048, 4096, 8192, 16384};
    for (int g : grids) {
      char name[64];
      snprintf(name, sizeof name, "copy 1 GiB, 4 in flight, grid %d", g);
      time(name, [&] {hipLaunchKernelGGL(copy_f4<4>, dim3(g), dim3(256), 0, 0, s4, d4, cn);}, 2.0 * cb / 1e9);
      snprintf(name, sizeof name, "copy 1 GiB, 8 in flight, grid %d", g);
      time(name, [&] {hipLaunchKernelGGL(copy_f4<8>, dim3(g), dim3(256), 0, 0, s4, d4, cn);}, 2.0 * cb / 1e9);
      snprintf(name, sizeof name, "copy 1 GiB, 1 in flight, grid %d", 16 * g);
      time(name, [&] {hipLaunchKernelGGL(copy_f4<1>, dim3(16 * g), dim3(256), 0, 0, s4, d4, cn);}, 2.0 * cb / 1e9);
    }
    hipMemset(pts, 1, bytes);
  }
  time("linear (grid 8192)", [&] {hipLaunchKernelGGL(read_linear, dim3(8192), dim3(256), 0, 0, pts, out, n);}, gb);
  time("4 rings per workgroup", [&] {hipLaunchKernelGGL(read_groups<4>, dim3(16 * 6, scans), dim3(256), 0, 0, pts, out);}, gb * 320 * 6 / 1800);
  time("4 rings per workgroup, XCD-local", [&] {hipLaunchKernelGGL((read_groups<4, 1>), dim3(16 * 6, scans), dim3(256), 0, 0, pts, out);}, gb * 320 * 6 / 1800);
  time("8 rings per workgroup", [&] {hipLaunchKernelGGL(read_groups<8>, dim3(8 * 12, scans), dim3(256), 0, 0, pts, out);}, gb * 160 * 12 / 1800);
  time("16 rings per workgroup", [&] {hipLaunchKernelGGL(read_groups<16>, dim3(4 * 23, scans), dim3(256), 0, 0, pts, out);}, gb * 80 * 23 / 1800);
  time("64 rings per workgroup", [&] {hipLaunchKernelGGL(read_groups<64>, dim3(1 * 90, scans), dim3(256), 0, 0, pts, out);}, gb * 20 * 90 / 1800);
  const double wgb = (double)scans * R * 1800 * 9 / 1e9;
  time("outputs: 1 B + 8 B per point", [&] {hipLaunchKernelGGL(write_outputs, dim3(16 * 6, scans), dim3(256), 0, 0, lab, cur, 0, pts);}, wgb);
  {
    float2 * sxy; float * sz; uint32_t * sidx; float4 * sp;
    const size_t rc = (size_t)scans * R * 1856;
    hipMalloc(&sxy, rc * 8); hipMalloc(&sz, rc * 4); hipMalloc(&sidx, rc * 4); hipMalloc(&sp, rc * 16);
    const double sgb = gb + (double)scans * R * C * 16 / 1e9;
    time("bucketing model: three staging arrays", [&] {hipLaunchKernelGGL(scatter_model<0>, dim3((C + 31) / 32, scans), dim3(256), 0, 0, pts, sxy, sz, sidx, sp);}, sgb);
    time("bucketing model: one 16-byte record array", [&] {hipLaunchKernelGGL(scatter_model<1>, dim3((C + 31) / 32, scans), dim3(256), 0, 0, pts, sxy, sz, sidx, sp);}, sgb);
    hipFree(sxy); hipFree(sz); hipFree(sidx); hipFree(sp);
  }
  {
    float4 * slots; float4 * po; uint32_t * io;
    const size_t n_slots = (size_t)scans * R * 6;
    hipMalloc(&slots, n_slots * 1280); hipMalloc(&po, (size_t)scans * R * 200 * 16); hipMalloc(&io, (size_t)scans * R * 200 * 4);
    hipMemset(slots, 0, n_slots * 1280);
    time("compaction model: 33 of 64 records per slot, wave per ring", [&] {hipLaunchKernelGGL(compact_model<64>, dim3(R / 4, scans), dim3(256), 0, 0, slots, po, io);},
      (double)n_slots * 33 * 40 / 1e9);
    time("compaction model: ... indices right behind the points", [&] {hipLaunchKernelGGL(compact_model<33>, dim3(R / 4, scans), dim3(256), 0, 0, slots, po, io);},
      (double)n_slots * 33 * 40 / 1e9);
    hipFree(slots); hipFree(po); hipFree(io);
  }
  // the unit kernel's memory side alone: what is left when the computation between loads and stores shrinks
  {
    float4 * rec; uint32_t * idx;
    hipMalloc(&rec, (size_t)scans * R * 1856 * 32); hipMalloc(&idx, (size_t)scans * R * 1856 * 4);
    const double rgb = gb * 320 * 6 / 1800, lgb = (double)scans * R * 1800 / 1e9, cgb = 8 * lgb, fgb = (double)scans * R * 6 * 38 * 20 / 1e9;
    const double ugb = rgb + lgb + cgb + fgb;
    const int works[] = {0, 64, 96, 128};
    const int ldss[] = {18000, 22144, 26000};       // 8 / 7 / 6 workgroups per CU
    for (int lds : ldss) {
      for (int work : works) {
        char name[96];
        snprintf(name, sizeof name, "unit memory, lds %d, work %d x16 fma", lds, work);
        time(name, [&] {hipLaunchKernelGGL(unit_memory<2>, dim3(16, 6, scans), dim3(256), lds, 0, pts, lab, cur, rec, idx, out, work);}, ugb);
      }
    }
#define UM(NAME, MAPV, GV, OUTV, LDS, GBV) \
    time(NAME, [&] {hipLaunchKernelGGL((unit_memory<MAPV, GV, OUTV>), dim3(R / GV, 6, scans), dim3(64 * GV), LDS, 0, pts, lab, cur, rec, idx, out, 64);}, GBV)
    UM("plain map, work 64", 0, 4, 7, 22144, ugb);
    UM("turned, work 64", 1, 4, 7, 22144, ugb);
    UM("pairs: no outputs at all", 2, 4, 0, 22144, rgb);
    UM("pairs: labels only", 2, 4, 1, 22144, rgb + lgb);
    UM("pairs: curvature only", 2, 4, 2, 22144, rgb + cgb);
    UM("pairs: records only", 2, 4, 4, 22144, rgb + fgb);
    UM("pairs: labels + curvature", 2, 4, 3, 22144, rgb + lgb + cgb);
    UM("pairs: labels + records (no curvature)", 2, 4, 5, 22144, rgb + lgb + fgb);
    UM("pairs: l + c + records as 32 B per feature", 2, 4, 11, 22144, ugb);
    UM("pairs: l + c + records by one store per array", 2, 4, 19, 22144, ugb);
    UM("pairs: l + c + 32 B records by one store", 2, 4, 35, 22144, ugb);
    UM("pairs: l + c + 32 B records by one store, slots of 64", 2, 4, 259, 22144, ugb);
    UM("pairs: l + c + 20 B records packed, three dword stores", 2, 4, 515, 22144, ugb);
    UM("pairs: l + c + 32 B records, non-temporal dwords", 2, 4, 1027, 22144, ugb);
    UM("pairs: l + c + 32 B records chunk by chunk, slots of 64", 2, 4, 2051, 22144, ugb);
    UM("pairs: l + c + records (again)", 2, 4, 7, 22144, ugb);
    UM("pairs: l + c + 32 B records by one store (again)", 2, 4, 35, 22144, ugb);
    UM("pairs: l + c + slots, indices right behind the points (as built)", 2, 4, 32771, 22144, ugb);
    UM("an XCD takes 12 consecutive (group, block) pairs, turned by the scan: same outputs", 3, 4, 32771, 22144, ugb);
    UM("an XCD takes 12 consecutive (group, block) pairs, not turned: same outputs", 4, 4, 32771, 22144, ugb);
    UM("pairs: no outputs (again)", 2, 4, 0, 22144, rgb);
    UM("12 consecutive, turned: no outputs", 3, 4, 0, 22144, rgb);
    UM("pairs: l + c + slots as built (20-byte planes)", 2, 4, 4099, 22144, ugb);
    UM("pairs: l + c + slots + the second leg inside (slot of 32 scans ago -> dense)", 2, 4, 12291, 22144, ugb + 2 * fgb);
    UM("pairs: l + c + records straight into dense clouds", 2, 4, 16387, 22144, ugb);
    UM("pairs: l + c + slots, indices right behind the points", 2, 4, 32771, 22144, ugb);
    UM("pairs: l + c + slots of 48, indices right behind the points", 2, 4, 65539, 22144, ugb);
    UM("pairs: l + c + slots as built (again)", 2, 4, 4099, 22144, ugb);
    UM("pairs: l + c + slots, indices right behind the points (again)", 2, 4, 32771, 22144, ugb);
    UM("pairs: l + c + slots + the second leg inside (again)", 2, 4, 12291, 22144, ugb + 2 * fgb);
    UM("8 rings per workgroup, turned, all outputs", 2, 8, 7, 44288, ugb);
    UM("16 rings per workgroup, turned, all outputs", 2, 16, 7, 88576, ugb);
    UM("8 rings per workgroup, no outputs", 2, 8, 0, 44288, rgb);
    UM("16 rings per workgroup, no outputs", 2, 16, 0, 88576, rgb);
  }
  return 0;
}
