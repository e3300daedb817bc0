// fetch_calib.hip — developer aid: what do rocprofv3's FETCH_SIZE / WRITE_SIZE report for the access widths of the step kernels?
// MI355X_MICROARCH.md calibrates FETCH_SIZE for 16-byte-per-lane streaming reads only (it reports half the bytes) and calls every
// other width and WRITE_SIZE uncalibrated; the step kernels load and store 4 (Ant) and 4 / 8 (Point) bytes per lane.  This program
// streams a buffer well past the 256 MiB Infinity Cache with one kernel per width, so that
//     rocprofv3 --pmc FETCH_SIZE -- ./fetch_calib      and      rocprofv3 --pmc WRITE_SIZE -- ./fetch_calib
// give counter / known-bytes factors per width (tools/fetch_calib.sh builds it on the GPU box, runs both passes and prints the table).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

template <class T>
__global__ void read_kernel(const T* __restrict__ src, size_t n, T* __restrict__ sink) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  T acc{};
  for (; i < n; i += stride) {
    T v = src[i];
    const uint32_t* w = reinterpret_cast<const uint32_t*>(&v);
    uint32_t* a = reinterpret_cast<uint32_t*>(&acc);
    for (unsigned k = 0; k < sizeof(T) / 4; k++) a[k] ^= w[k];
  }
  const uint32_t* a = reinterpret_cast<const uint32_t*>(&acc);
  uint32_t x = 0;
  for (unsigned k = 0; k < sizeof(T) / 4; k++) x ^= a[k];
  if (x == 0x12345678u) sink[0] = acc;  // never true for the zero-filled buffer's pattern: keeps the loads alive
}
template <class T>
__global__ void write_kernel(T* __restrict__ dst, size_t n, uint32_t seed) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    T v;
    uint32_t* w = reinterpret_cast<uint32_t*>(&v);
    for (unsigned k = 0; k < sizeof(T) / 4; k++) w[k] = seed + (uint32_t)i;
    dst[i] = v;
  }
}
// the Ant kernel's record pattern: 16 lanes read 48 consecutive words of one 192-byte record (three 4-byte loads per lane)
__global__ void record_kernel(const float* __restrict__ src, size_t nrec, float* __restrict__ sink) {
  size_t g = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
  const int l = threadIdx.x & 15;
  float acc = 0.f;
  for (; g < nrec; g += ((size_t)gridDim.x * blockDim.x) >> 4) acc += src[g * 48 + l] + src[g * 48 + 16 + l] + src[g * 48 + 32 + l];
  if (acc == 1.2345f) sink[0] = acc;
}

// instruction fetch: 128 KiB of straight-line code (32768 four-byte VALU instructions), run by 8 / 64 / 1024 one-wave workgroups — what
// does FETCH_SIZE count per launch for CODE (the step kernels' fixed fetch term: every XCD's L2 fetches the code it executes)?
template <int TAG>
__global__ void code_kernel(unsigned* sink) {
  unsigned x = threadIdx.x;
  asm volatile(".rept 32768\n\tv_add_u32_e32 %0, 1, %0\n\t.endr" : "+v"(x));
  if (x == 0xdeadbeefu) sink[0] = x + TAG;
}

int main() {
  const size_t bytes = (size_t)1 << 30;
  void *buf, *sink;
  hipMalloc(&buf, bytes); hipMalloc(&sink, 64);
  hipMemset(buf, 1, bytes);
  hipDeviceSynchronize();
  const dim3 grid(256 * 16), blk(256);
  for (int rep = 0; rep < 3; rep++) {
    hipLaunchKernelGGL(read_kernel<uint32_t>, grid, blk, 0, 0, (const uint32_t*)buf, bytes / 4, (uint32_t*)sink);
    hipLaunchKernelGGL(read_kernel<uint2>, grid, blk, 0, 0, (const uint2*)buf, bytes / 8, (uint2*)sink);
    hipLaunchKernelGGL(read_kernel<uint4>, grid, blk, 0, 0, (const uint4*)buf, bytes / 16, (uint4*)sink);
    hipLaunchKernelGGL(record_kernel, grid, blk, 0, 0, (const float*)buf, bytes / 192, (float*)sink);
    hipLaunchKernelGGL(write_kernel<uint32_t>, grid, blk, 0, 0, (uint32_t*)buf, bytes / 4, 7u);
    hipLaunchKernelGGL(write_kernel<uint2>, grid, blk, 0, 0, (uint2*)buf, bytes / 8, 7u);
    hipLaunchKernelGGL(write_kernel<uint4>, grid, blk, 0, 0, (uint4*)buf, bytes / 16, 7u);
  }
  hipDeviceSynchronize();
  for (int rep = 0; rep < 3; rep++) {  // distinct instantiations = distinct code addresses: every launch fetches code nobody has cached
    if (rep == 0) { hipLaunchKernelGGL(code_kernel<0>, dim3(8), dim3(64), 0, 0, (unsigned*)sink); hipLaunchKernelGGL(code_kernel<1>, dim3(64), dim3(64), 0, 0, (unsigned*)sink); hipLaunchKernelGGL(code_kernel<2>, dim3(1024), dim3(64), 0, 0, (unsigned*)sink); }
    if (rep == 1) { hipLaunchKernelGGL(code_kernel<0>, dim3(8), dim3(64), 0, 0, (unsigned*)sink); hipLaunchKernelGGL(code_kernel<1>, dim3(64), dim3(64), 0, 0, (unsigned*)sink); hipLaunchKernelGGL(code_kernel<2>, dim3(1024), dim3(64), 0, 0, (unsigned*)sink); }
    if (rep == 2) { hipMemset(buf, 2, bytes); hipLaunchKernelGGL(code_kernel<2>, dim3(1024), dim3(64), 0, 0, (unsigned*)sink); }  // after 1 GiB of other traffic
    hipDeviceSynchronize();
  }
  printf("fetch_calib: %zu bytes per kernel, 3 repetitions\n", bytes);
  return 0;
}
