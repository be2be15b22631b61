// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for the access shapes of this repo's kernels (MI355X_MICROARCH.md, HBM:
// "other access widths are uncalibrated: calibrate on a known byte count in your own access pattern").  Each kernel moves a
// known number of bytes over a 1 GiB buffer (larger than the 256 MiB Infinity Cache); run under
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- ./pmc_calibrate
//   rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out -- ./pmc_calibrate
// and divide the counter (KB) by the byte count printed here.   hipcc --offload-arch=gfx950 -O2 tools/pmc_calibrate.hip -o pmc_calibrate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void rd16(const uint4* __restrict__ a, size_t n, uint32_t* sink) {          // 16 B per lane, coalesced
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { uint4 v = a[i]; acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    if (acc == 0x12345u) *sink = acc;
}
__global__ void rd4(const uint32_t* __restrict__ a, size_t n, uint32_t* sink) {        // 4 B per lane, coalesced
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= a[i];
    if (acc == 0x12345u) *sink = acc;
}
// 4 B per lane, lane = row, rows of RW words read word by word (the k-mer kernel's read-row loads)
template <int RW>
__global__ void rdrow(const uint32_t* __restrict__ a, size_t rows, uint32_t* sink) {
    uint32_t acc = 0;
    for (size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x; r < rows; r += (size_t)gridDim.x * blockDim.x)
#pragma unroll
        for (int w = 0; w < RW; w++) acc ^= a[r * RW + w];
    if (acc == 0x12345u) *sink = acc;
}
__global__ void wr16(uint4* a, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = make_uint4(i, 1, 2, 3);
}
__global__ void wr4(uint32_t* a, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = (uint32_t)i;
}
// one 4-byte store every `stride` words per lane (the scattered per-unique-read stores the k-mer kernel used to make)
__global__ void wr4_strided(uint32_t* a, size_t n, int stride) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i * stride < n; i += (size_t)gridDim.x * blockDim.x) a[i * stride] = (uint32_t)i;
}
// one byte per lane, coalesced (flag arrays)
__global__ void wr1(uint8_t* a, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = (uint8_t)i;
}
// LDS-free global atomicAdd, one dword per lane, distinct addresses
__global__ void at4(uint32_t* a, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) atomicAdd(&a[i], 1u);
}

int main() {
    const size_t bytes = (size_t)1 << 30;
    void* buf; uint32_t* sink;
    CHECK(hipMalloc(&buf, bytes)); CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(buf, 1, bytes));
    const int g = 256 * 8, b = 256;
    rd16<<<g, b>>>((const uint4*)buf, bytes / 16, sink);                 printf("rd16 bytes %zu\n", bytes);
    rd4<<<g, b>>>((const uint32_t*)buf, bytes / 4, sink);                printf("rd4 bytes %zu\n", bytes);
    rdrow<7><<<g, b>>>((const uint32_t*)buf, bytes / 28, sink);          printf("rdrow<7> bytes %zu\n", bytes / 28 * 28);
    rdrow<11><<<g, b>>>((const uint32_t*)buf, bytes / 44, sink);         printf("rdrow<11> bytes %zu\n", bytes / 44 * 44);
    wr16<<<g, b>>>((uint4*)buf, bytes / 16);                             printf("wr16 bytes %zu\n", bytes);
    wr4<<<g, b>>>((uint32_t*)buf, bytes / 4);                            printf("wr4 bytes %zu\n", bytes);
    wr4_strided<<<g, b>>>((uint32_t*)buf, bytes / 4, 6);                 printf("wr4_strided(6) bytes %zu\n", bytes / 24 * 4);
    wr1<<<g, b>>>((uint8_t*)buf, bytes / 4);                             printf("wr1 bytes %zu\n", bytes / 4);
    at4<<<g, b>>>((uint32_t*)buf, bytes / 16);                           printf("at4 bytes %zu\n", bytes / 4);
    CHECK(hipDeviceSynchronize());
    return 0;
}
