// The reduction ledger (reduce_ledger.h) and its C entry points: lt_reduce_begin / lt_reduce_flush / lt_reduce_end.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "lt_common.h"
#include "reduce_ledger.h"

namespace {

struct Ent { const float* src; float* dst; int nparts; int C; long stride; };
struct DevEnt { const float* src; long stride; int nparts; int pad; };
struct DevGrp { float* dst; int C; int first; int count; int vec; };

std::mutex mu;
float* scratch = nullptr;
size_t cap = 0, used = 0;
bool on = false;
std::vector<Ent> pending;
long overflows = 0;
int flush_idx = 0;
constexpr int PIN_RING = 8;
struct Slot {
  std::vector<char> host; void* dev = nullptr; size_t dev_cap = 0; void* pinned = nullptr; size_t pinned_cap = 0;
  // eager uploads: a ring of pinned staging buffers, each guarded by the event recorded behind its last copy (round 6)
  void* pin[PIN_RING] = {nullptr}; size_t pin_cap[PIN_RING] = {0}; hipEvent_t pin_ev[PIN_RING] = {nullptr}; unsigned next = 0;
};
std::vector<Slot> slots;   // one cached device table per flush of a step: identical steps upload nothing

// One workgroup per (destination, 256-column group): 64 column lanes x 4 columns, 4 part lanes.  Part lane q adds the partial rows
// q, q + 4, ... of every entry in order; the four lane sums are combined in LDS in the order ((0 + 1) + 2) + 3.
__global__ __launch_bounds__(256) void ledger_reduce_kernel(const DevGrp* __restrict__ grps, const DevEnt* __restrict__ ents) {
  __shared__ float4 red[4][64];
  const DevGrp g = grps[blockIdx.y];
  const int cl = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int c = (blockIdx.x * 64 + cl) * 4;
  if (blockIdx.x * 256 >= g.C) return;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < g.C) {
    for (int e = g.first; e < g.first + g.count; ++e) {
      const DevEnt en = ents[e];
      const float* s = en.src + c;
      if (g.vec) {
        int p = q;
        for (; p + 12 < en.nparts; p += 16) {   // four independent loads in flight
          const float4 v0 = *reinterpret_cast<const float4*>(s + (long)p * en.stride);
          const float4 v1 = *reinterpret_cast<const float4*>(s + (long)(p + 4) * en.stride);
          const float4 v2 = *reinterpret_cast<const float4*>(s + (long)(p + 8) * en.stride);
          const float4 v3 = *reinterpret_cast<const float4*>(s + (long)(p + 12) * en.stride);
          acc.x += v0.x; acc.y += v0.y; acc.z += v0.z; acc.w += v0.w;
          acc.x += v1.x; acc.y += v1.y; acc.z += v1.z; acc.w += v1.w;
          acc.x += v2.x; acc.y += v2.y; acc.z += v2.z; acc.w += v2.w;
          acc.x += v3.x; acc.y += v3.y; acc.z += v3.z; acc.w += v3.w;
        }
        for (; p < en.nparts; p += 4) {
          const float4 v0 = *reinterpret_cast<const float4*>(s + (long)p * en.stride);
          acc.x += v0.x; acc.y += v0.y; acc.z += v0.z; acc.w += v0.w;
        }
      } else {
        for (int p = q; p < en.nparts; p += 4) {
          const float* r = s + (long)p * en.stride;
          acc.x += r[0];
          if (c + 1 < g.C) acc.y += r[1];
          if (c + 2 < g.C) acc.z += r[2];
          if (c + 3 < g.C) acc.w += r[3];
        }
      }
    }
  }
  red[q][cl] = acc;
  __syncthreads();
  if (q == 0 && c < g.C) {
    const float4 a = red[0][cl], b = red[1][cl], d = red[2][cl], e = red[3][cl];
    float* o = g.dst + c;
    o[0] += ((a.x + b.x) + d.x) + e.x;
    if (c + 1 < g.C) o[1] += ((a.y + b.y) + d.y) + e.y;
    if (c + 2 < g.C) o[2] += ((a.z + b.z) + d.z) + e.z;
    if (c + 3 < g.C) o[3] += ((a.w + b.w) + d.w) + e.w;
  }
}

int flush_locked(hipStream_t st) {
  if (pending.empty()) return LT_OK;
  // group by destination, keeping record order inside a group and first-seen order between groups
  std::unordered_map<float*, int> gi;
  std::vector<std::vector<int>> members;
  for (int i = 0; i < (int)pending.size(); ++i) {
    auto it = gi.find(pending[i].dst);
    if (it == gi.end()) { gi.emplace(pending[i].dst, (int)members.size()); members.push_back({i}); }
    else members[it->second].push_back(i);
  }
  const size_t ng = members.size(), ne = pending.size();
  std::vector<char> host(ng * sizeof(DevGrp) + ne * sizeof(DevEnt));
  DevGrp* hg = reinterpret_cast<DevGrp*>(host.data());
  DevEnt* he = reinterpret_cast<DevEnt*>(host.data() + ng * sizeof(DevGrp));
  int maxC = 0, k = 0;
  for (size_t g = 0; g < ng; ++g) {
    const Ent& f = pending[members[g][0]];
    bool vec = f.C % 4 == 0 && ((uintptr_t)f.dst & 15) == 0;
    hg[g] = DevGrp{f.dst, f.C, k, (int)members[g].size(), 0};
    for (int i : members[g]) {
      const Ent& e = pending[i];
      if (e.C != f.C) { lt_set_error("lt_reduce_flush: one destination recorded with two widths"); return LT_ERR_INVALID; }
      vec = vec && e.stride % 4 == 0 && ((uintptr_t)e.src & 15) == 0;
      he[k++] = DevEnt{e.src, e.stride, e.nparts, 0};
    }
    hg[g].vec = vec ? 1 : 0;
    maxC = f.C > maxC ? f.C : maxC;
  }
  if ((int)slots.size() <= flush_idx) slots.resize(flush_idx + 1);
  Slot& s = slots[flush_idx++];
  if (s.host != host) {
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(st, &cs);
    const bool capturing = cs != hipStreamCaptureStatusNone;
    if (s.dev_cap < host.size() || s.pinned_cap < host.size()) {
      // (allocation is not legal while a stream captures: a flush that is captured into a HIP graph must have run once, eagerly, with the
      // same table size before -- the step that precedes the capture does exactly that)
      if (capturing) { lt_set_error("lt_reduce_flush: slot %d has no table buffers yet and the stream is capturing", flush_idx - 1); return LT_ERR_INVALID; }
      if (s.dev) hipFree(s.dev);
      if (s.pinned) hipHostFree(s.pinned);
      s.dev_cap = s.pinned_cap = host.size() * 2;
      if (hipMalloc(&s.dev, s.dev_cap) != hipSuccess || hipHostMalloc(&s.pinned, s.pinned_cap, hipHostMallocDefault) != hipSuccess) {
        s.dev = s.pinned = nullptr; s.dev_cap = s.pinned_cap = 0; lt_set_error("lt_reduce_flush: table allocation failed"); return LT_ERR_HIP;
      }
    }
    // Under stream capture the copy becomes a graph node that re-reads its source at every replay: it goes through the slot's own pinned
    // buffer, written here and left alone afterwards (a slot range that is captured belongs to the graph's region: lt_reduce_begin_at).
    // Eagerly the table goes through a ring of pinned staging buffers (round 6).  It used to be uploaded from pageable memory, which the
    // runtime stages before returning -- by waiting for the stream to reach the copy: a table that changes every step (the first ledger
    // region's does: its partial counts follow the number of masked tokens) cost the launch thread its whole lead over the device at
    // every flush, and with one flush per region that was a bubble per region boundary.  A pinned source is a queued DMA; a ring entry is
    // taken again only after the event recorded behind its last copy (8 deep: further than the launch thread ever leads).
    if (capturing) {
      memcpy(s.pinned, host.data(), host.size());
      if (hipMemcpyAsync(s.dev, s.pinned, host.size(), hipMemcpyHostToDevice, st) != hipSuccess) { lt_set_error("lt_reduce_flush: table upload failed"); return LT_ERR_HIP; }
    } else {
      const unsigned r = s.next++ % PIN_RING;
      if (s.pin_ev[r]) (void)hipEventSynchronize(s.pin_ev[r]);
      if (s.pin_cap[r] < host.size()) {
        if (s.pin[r]) hipHostFree(s.pin[r]);
        s.pin_cap[r] = host.size() * 2;
        if (hipHostMalloc(&s.pin[r], s.pin_cap[r], hipHostMallocDefault) != hipSuccess) {
          s.pin[r] = nullptr; s.pin_cap[r] = 0; lt_set_error("lt_reduce_flush: pinned staging allocation failed"); return LT_ERR_HIP;
        }
      }
      if (!s.pin_ev[r] && hipEventCreateWithFlags(&s.pin_ev[r], hipEventDisableTiming) != hipSuccess) { lt_set_error("lt_reduce_flush: event creation failed"); return LT_ERR_HIP; }
      memcpy(s.pin[r], host.data(), host.size());
      if (hipMemcpyAsync(s.dev, s.pin[r], host.size(), hipMemcpyHostToDevice, st) != hipSuccess) { lt_set_error("lt_reduce_flush: table upload failed"); return LT_ERR_HIP; }
      (void)hipEventRecord(s.pin_ev[r], st);
    }
    s.host = host;
  }
  const DevGrp* dg = reinterpret_cast<const DevGrp*>(s.dev);
  const DevEnt* de = reinterpret_cast<const DevEnt*>((const char*)s.dev + ng * sizeof(DevGrp));
  hipLaunchKernelGGL(ledger_reduce_kernel, dim3((unsigned)lt_cdiv(maxC, 256), (unsigned)ng), dim3(256), 0, st, dg, de);
  pending.clear();
  if (hipGetLastError() != hipSuccess) { lt_set_error("lt_reduce_flush: launch failed"); return LT_ERR_HIP; }
  return LT_OK;
}

}  // namespace

float* lt_scratch_ring(size_t floats) {
  static std::mutex rmu;
  static float* buf[8] = {nullptr};
  static size_t bcap[8] = {0};
  static unsigned next = 0;
  std::lock_guard<std::mutex> l(rmu);
  const unsigned i = next++ % 8;
  if (bcap[i] < floats) {
    // grow: the old buffer may still be read by a kernel in flight, so it is left allocated (a few KiB .. MiB, a handful of times per process)
    float* p = nullptr;
    const size_t want = floats < 65536 ? 65536 : floats * 2;
    if (hipMalloc(&p, want * sizeof(float)) != hipSuccess) return nullptr;
    buf[i] = p; bcap[i] = want;
  }
  return buf[i];
}

namespace lt_ledger {
bool active() { std::lock_guard<std::mutex> l(mu); return on; }
float* reserve(size_t floats) {
  std::lock_guard<std::mutex> l(mu);
  if (!on) return nullptr;
  floats = (floats + 3) & ~(size_t)3;
  if (used + floats > cap) { ++overflows; return nullptr; }
  float* p = scratch + used;
  used += floats;
  return p;
}
void record(float* dst, const float* src, int nparts, long stride, int C) {
  std::lock_guard<std::mutex> l(mu);
  pending.push_back(Ent{src, dst, nparts, C, stride});
}
}  // namespace lt_ledger

extern "C" int lt_reduce_begin_at(float* scratch_f32, int64_t floats, int first_slot) {
  LT_CHECK_ARG(scratch_f32 && floats > 0 && ((uintptr_t)scratch_f32 & 15) == 0 && first_slot >= 0 && first_slot < 4096,
               "lt_reduce_begin: scratch must be a 16-byte aligned region (and 0 <= first_slot < 4096)");
  std::lock_guard<std::mutex> l(mu);
  pending.clear();   // entries of a step that was abandoned half-way (its gradients are discarded with it)
  scratch = scratch_f32; cap = (size_t)floats; used = 0; on = true; flush_idx = first_slot;
  return LT_OK;
}
extern "C" int lt_reduce_begin(float* scratch_f32, int64_t floats) { return lt_reduce_begin_at(scratch_f32, floats, 0); }
extern "C" int lt_reduce_flush(void* stream) {
  std::lock_guard<std::mutex> l(mu);
  return flush_locked((hipStream_t)stream);
}
extern "C" int lt_reduce_end(void* stream) {
  std::lock_guard<std::mutex> l(mu);
  const int rc = flush_locked((hipStream_t)stream);
  on = false;
  return rc;
}
extern "C" int64_t lt_reduce_overflows(void) {
  std::lock_guard<std::mutex> l(mu);
  return overflows;
}
