#include "segsde_common.h"
extern "C" int segsde_abi_version(void) { return SEGSDE_ABI_VERSION; }
#include <mutex>

// A zeroed slice of the device-resident ticket ring for one launch (n tiles).  Consecutive launches get disjoint slices, so
// launches that overlap in time never share a ticket; every slice is left zeroed by the launch that used it.
unsigned* segsde_ticket_slice(int n) {
  constexpr size_t CAP = 1u << 20;
  static std::mutex mu;
  static unsigned* buf[64] = {};
  static size_t pos[64] = {};
  std::lock_guard<std::mutex> lock(mu);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64 || (size_t)n > CAP) return nullptr;
  if (!buf[dev]) {
    void* q = nullptr;
    if (hipMalloc(&q, CAP * sizeof(unsigned)) != hipSuccess) return nullptr;
    if (hipMemset(q, 0, CAP * sizeof(unsigned)) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return nullptr;
    buf[dev] = static_cast<unsigned*>(q);
  }
  if (pos[dev] + n > CAP) pos[dev] = 0;
  unsigned* r = buf[dev] + pos[dev];
  pos[dev] += n;
  return r;
}
