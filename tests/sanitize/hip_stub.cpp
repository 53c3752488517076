// Test infrastructure (CPU box only): a stand-in for the 33 HIP runtime entry points libbnmtf_hip.so imports, over plain host
// memory, so that the HOST side of the library -- layout passes on worker threads, slot / hand-over table builders, arena and
// pinned-buffer pools, the in-process multi-rank rendezvous of comm.hip, every H2D / D2H copy's extent -- runs under
// AddressSanitizer / UndefinedBehaviorSanitizer / ThreadSanitizer without a GPU (`make asan`, `make tsan`).
// "Device" memory is calloc'ed host memory (so a copy that overruns a device buffer IS a heap overflow the sanitizer sees),
// kernel launches do nothing, streams are synchronous.  Never linked into the product.
#include <hip/hip_runtime.h>

#include <atomic>
#include <stdexcept>
#include <chrono>
#include <cstdlib>
#include <cstring>

namespace {
thread_local int t_device = 0;
std::atomic<long> g_launches{0}, g_allocs{0};
struct StubEvent { std::chrono::steady_clock::time_point t; };
int device_count() { const char* e = getenv("HIPSTUB_DEVICES"); return e ? atoi(e) : 1; }
}  // namespace

extern "C" {
long hipstub_launches() { return g_launches.load(); }
long hipstub_live_allocs() { return g_allocs.load(); }

void** __hipRegisterFatBinary(const void*) { static void* h = nullptr; return &h; }
void __hipUnregisterFatBinary(void**) {}
void __hipRegisterFunction(void**, const void*, char*, const char*, unsigned, void*, void*, void*, void*, int*) {}
void __hipRegisterVar(void**, void*, char*, const char*, int, size_t, int, int) {}
hipError_t __hipPushCallConfiguration(dim3, dim3, size_t, hipStream_t) { return hipSuccess; }
hipError_t __hipPopCallConfiguration(dim3*, dim3*, size_t*, hipStream_t*) { return hipSuccess; }
hipError_t hipLaunchKernel(const void*, dim3, dim3, void**, size_t, hipStream_t) { ++g_launches; return hipSuccess; }

hipError_t hipGetDeviceCount(int* n) { *n = device_count(); return hipSuccess; }
hipError_t hipSetDevice(int d) { if (d < 0 || d >= device_count()) return hipErrorInvalidDevice; t_device = d; return hipSuccess; }
hipError_t hipGetDevice(int* d) { *d = t_device; return hipSuccess; }
hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipGetLastError() { return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "hip_stub"; }
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }

hipError_t hipMalloc(void** p, size_t n) { *p = calloc(n ? n : 1, 1); if (!*p) return hipErrorOutOfMemory; ++g_allocs; return hipSuccess; }
hipError_t hipFree(void* p) { if (p) { free(p); --g_allocs; } return hipSuccess; }
hipError_t hipHostMalloc(void** p, size_t n, unsigned) { *p = calloc(n ? n : 1, 1); if (!*p) return hipErrorOutOfMemory; ++g_allocs; return hipSuccess; }
hipError_t hipHostFree(void* p) { if (p) { free(p); --g_allocs; } return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { if (n) memmove(d, s, n); return hipSuccess; }
hipError_t hipMemset(void* d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { if (n) memset(d, v, n); return hipSuccess; }
hipError_t hipPointerGetAttributes(hipPointerAttribute_t*, const void*) { return hipErrorInvalidValue; }      // "not a registered pointer": pageable

hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = reinterpret_cast<hipStream_t>(malloc(8)); ++g_allocs; return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { free(s); --g_allocs; return hipSuccess; }
// hipstub_throw_at_sync(n): the n-th stream synchronisation from now throws -- a C++ exception inside an entry point of the library,
// which must come back as a status, not cross the C ABI (api.hip: BNMTF_ABI_GUARD)
static thread_local long t_throw_at = 0;          // (per host thread: the driver walks the ABI from several at a time)
extern "C" void hipstub_throw_at_sync(long n) { t_throw_at = n; }
hipError_t hipStreamSynchronize(hipStream_t) {
  if (t_throw_at > 0 && --t_throw_at == 0) throw std::runtime_error("injected by the stub");
  return hipSuccess;
}
hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = reinterpret_cast<hipEvent_t>(new StubEvent{std::chrono::steady_clock::now()}); ++g_allocs; return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { delete reinterpret_cast<StubEvent*>(e); --g_allocs; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { reinterpret_cast<StubEvent*>(e)->t = std::chrono::steady_clock::now(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
  *ms = std::chrono::duration<float, std::milli>(reinterpret_cast<StubEvent*>(b)->t - reinterpret_cast<StubEvent*>(a)->t).count();
  return hipSuccess;
}
}
