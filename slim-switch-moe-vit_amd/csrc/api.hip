// C-ABI bookkeeping: version + thread-local error message.
#include "smoe_common.h"
#include <string.h>

static thread_local char g_err[512] = "";

void smoe_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

// ---- kernel registry (smoe_common.h) ---------------------------------------------------------------------------------
static std::atomic<SmoeKernelEntry*>& registry_head() {
  static std::atomic<SmoeKernelEntry*> head{nullptr};
  return head;
}
SmoeKernelEntry::SmoeKernelEntry(const void* k) : kern(k), done(0), next(nullptr) {
  auto& head = registry_head();
  next = head.load(std::memory_order_relaxed);
  while (!head.compare_exchange_weak(next, this, std::memory_order_release, std::memory_order_relaxed)) {}
}
static constexpr int SMOE_MAX_LDS = 160 * 1024;  // the whole LDS of a gfx950 CU; the attribute is only a cap
int smoe_kernel_ensure(SmoeKernelEntry& e) {
  int dev = 0;
  hipError_t ge = hipGetDevice(&dev);
  if (ge != hipSuccess) {
    smoe_set_error("hipGetDevice failed: %s", hipGetErrorString(ge));
    return (int)ge;
  }
  const uint32_t bit = (dev >= 0 && dev < 32) ? (1u << dev) : 0u;
  if (bit && (e.done.load(std::memory_order_acquire) & bit)) return 0;
  hipError_t ae = hipFuncSetAttribute(e.kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMOE_MAX_LDS);
  if (ae != hipSuccess) {
    smoe_set_error("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed on device %d: %s", dev, hipGetErrorString(ae));
    return (int)ae;
  }
  if (bit) e.done.fetch_or(bit, std::memory_order_release);
  return 0;
}
namespace {
__global__ void zero_words_kernel(uint32_t* p, int64_t words) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (int64_t)gridDim.x * blockDim.x) p[i] = 0u;
}
}  // namespace

hipError_t smoe_zero_words(void* p, int64_t words, hipStream_t s) {
  if (words <= 0) return hipSuccess;
  const int64_t blocks = (words + 255) / 256;
  hipLaunchKernelGGL(zero_words_kernel, dim3((unsigned)(blocks > 1024 ? 1024 : blocks)), dim3(256), 0, s, (uint32_t*)p, words);
  return hipGetLastError();
}

int smoe_num_cus() {
  static std::atomic<int> cached[32];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  if (dev >= 0 && dev < 32) {
    const int c = cached[dev].load(std::memory_order_relaxed);
    if (c > 0) return c;
  }
  int v = 0;
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
  if (dev >= 0 && dev < 32) cached[dev].store(v, std::memory_order_relaxed);
  return v;
}

extern "C" int smoe_init(void) {
  for (SmoeKernelEntry* e = registry_head().load(std::memory_order_acquire); e; e = e->next) {
    const int rc = smoe_kernel_ensure(*e);
    if (rc != 0) return rc;
  }
  (void)smoe_num_cus();
  return 0;
}

extern "C" int smoe_abi_version(void) { return 18; }
extern "C" const char* smoe_last_error(void) { return g_err; }
