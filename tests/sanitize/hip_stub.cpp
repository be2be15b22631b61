// TEST INFRASTRUCTURE: a stand-in for the HIP runtime so that the HOST half of libbreakmer_hip.so (bk_api.hip compiled with
// `hipcc --cuda-host-only`: packing threads, asynchronous submit, staging reuse, arena growth logic, getters, call tail) can
// run under AddressSanitizer / UndefinedBehaviorSanitizer / ThreadSanitizer on a box without a GPU (GPU-side sanitizers are
// not available on this pool).  "Device" memory is zeroed host memory, copies are memcpy, kernel launches do nothing:
// every region then reports zero k-mers and zero contigs, which is all the host code paths under test need.
#include <hip/hip_runtime_api.h>
#include <atomic>
#include <cstdlib>
#include <cstring>

static std::atomic<long> g_launches{0}, g_allocs{0};
extern "C" long bk_stub_launches() { return g_launches.load(); }
extern "C" long bk_stub_live_allocs() { return g_allocs.load(); }

extern "C" {
hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_tR0600 *p, int) { memset(p, 0, sizeof(*p)); strcpy(p->gcnArchName, "gfx950:sramecc+:xnack-"); p->multiProcessorCount = 256; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = (hipStream_t)calloc(1, 8); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t *e) { *e = (hipEvent_t)calloc(1, 8); return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float *ms, hipEvent_t, hipEvent_t) { *ms = 0.0f; return hipSuccess; }
hipError_t hipMalloc(void **p, size_t n) { *p = calloc(n ? n : 1, 1); if (*p) g_allocs++; return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void *p) { if (p) g_allocs--; free(p); return hipSuccess; }
hipError_t hipHostMalloc(void **p, size_t n, unsigned) { *p = calloc(n ? n : 1, 1); if (*p) g_allocs++; return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void *p) { if (p) g_allocs--; free(p); return hipSuccess; }
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
hipError_t hipMemset(void *d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
hipError_t hipFuncSetAttribute(const void *, hipFuncAttribute, int) { return hipSuccess; }
hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int *n, const void *, int, size_t) { *n = 2; return hipSuccess; }
hipError_t hipLaunchKernel(const void *, dim3, dim3, void **, size_t, hipStream_t) { g_launches++; return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
const char *hipGetErrorString(hipError_t) { return "stub"; }
// what the host-side launch stubs and the module constructor of a --cuda-host-only object call
void **__hipRegisterFatBinary(const void *) { static void *h; return &h; }
void __hipUnregisterFatBinary(void **) {}
void __hipRegisterFunction(void **, const void *, char *, const char *, unsigned, void *, void *, void *, void *, int *) {}
static thread_local struct { dim3 g, b; size_t shm; hipStream_t s; } g_cfg;
hipError_t __hipPushCallConfiguration(dim3 g, dim3 b, size_t shm, hipStream_t s) { g_cfg.g = g; g_cfg.b = b; g_cfg.shm = shm; g_cfg.s = s; return hipSuccess; }
hipError_t __hipPopCallConfiguration(dim3 *g, dim3 *b, size_t *shm, hipStream_t *s) { *g = g_cfg.g; *b = g_cfg.b; *shm = g_cfg.shm; *s = g_cfg.s; return hipSuccess; }
}
