// Error state, version and device probing for libnncf_hip.so.
#include <cstring>
#include <string>
#include <algorithm>
#include <vector>

#include "common.h"

namespace nncf {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int fail(int status, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return status;
}

namespace {
struct GlobalProf {
  std::string target;
  std::vector<hipEvent_t> start, stop;
  size_t used = 0;
  bool on = false;
  hipStream_t last = nullptr;
} g_prof;
}  // namespace

hipEvent_t* gprof_start(const char* kernel, hipStream_t stream) {
  if (!g_prof.on || g_prof.used >= g_prof.start.size() || g_prof.target != kernel) return nullptr;
  g_prof.last = stream;
  (void)hipEventRecord(g_prof.start[g_prof.used], stream);
  return &g_prof.stop[g_prof.used++];
}

void gprof_stop(hipEvent_t* stop, hipStream_t stream) {
  if (stop) (void)hipEventRecord(*stop, stream);
}

}  // namespace nncf

extern "C" {

int nncf_profile_begin(const char* kernel, int64_t max_launches) {
  using namespace nncf;
  NNCF_REQUIRE(kernel && max_launches > 0 && max_launches <= 65536, "bad argument");
  while ((int64_t)g_prof.start.size() < max_launches) {
    hipEvent_t a, b;
    NNCF_HIP_CHECK(hipEventCreate(&a));
    NNCF_HIP_CHECK(hipEventCreate(&b));
    g_prof.start.push_back(a);
    g_prof.stop.push_back(b);
  }
  g_prof.target = kernel;
  g_prof.used = 0;
  g_prof.on = true;
  return NNCF_OK;
}

int nncf_profile_end(double* avg_ms, int64_t* launches, double* overhead_ms) {
  using namespace nncf;
  NNCF_REQUIRE(avg_ms && launches && overhead_ms, "NULL argument");
  g_prof.on = false;
  double total = 0;
  for (size_t i = 0; i < g_prof.used; ++i) {
    NNCF_HIP_CHECK(hipEventSynchronize(g_prof.stop[i]));
    float ms = 0;
    NNCF_HIP_CHECK(hipEventElapsedTime(&ms, g_prof.start[i], g_prof.stop[i]));
    total += ms;
  }
  *launches = (int64_t)g_prof.used;
  *avg_ms = g_prof.used ? total / (double)g_prof.used : 0.0;
  // Calibration: the same event pair with nothing between them, on the same stream: the cost of
  // the two marker packets themselves, which every bracket above includes.
  const size_t ncal = g_prof.start.size() < 64 ? g_prof.start.size() : 64;
  NNCF_HIP_CHECK(hipStreamSynchronize(g_prof.last));
  for (size_t i = 0; i < ncal; ++i) {
    NNCF_HIP_CHECK(hipEventRecord(g_prof.start[i], g_prof.last));
    NNCF_HIP_CHECK(hipEventRecord(g_prof.stop[i], g_prof.last));
  }
  std::vector<float> cal(ncal, 0.f);
  for (size_t i = 0; i < ncal; ++i) {
    NNCF_HIP_CHECK(hipEventSynchronize(g_prof.stop[i]));
    NNCF_HIP_CHECK(hipEventElapsedTime(&cal[i], g_prof.start[i], g_prof.stop[i]));
  }
  std::sort(cal.begin(), cal.end());
  *overhead_ms = ncal ? (double)cal[ncal / 2] : 0.0;  // median: a stray slow pair must not shift it
  g_prof.used = 0;
  return NNCF_OK;
}


const char* nncf_last_error(void) { return nncf::g_err; }

const char* nncf_version(void) { return "nncf_hip 0.1 (gfx950)"; }

int nncf_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

}  // extern "C"
