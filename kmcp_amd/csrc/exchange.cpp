// exchange.cpp — the one exchange step of the path inside a single host process that drives several GPUs
// (kmcpg_open_devices: what a cgo host or `kmcp-search --gpus N` uses): every GPU has searched the batch against its share of
// the index blocks; the per-read hit lists are brought to the first GPU over xGMI with RCCL (grouped ncclSend / ncclRecv of
// exactly the bytes each shard produced) and leave the node's GPUs with ONE device-to-host copy.  The reference's counterpart is
// the concatenation of its per-block workers' replies (util-db-search.go:939-964).
//
// RCCL is bound at run time (dlopen of librccl.so.1: the copy PyTorch has already mapped in a Python host, ROCm's otherwise), so
// the library has no link-time dependency on it and hosts with one GPU never load it.  If RCCL cannot be initialised — duplicate
// device ordinals (RCCL refuses two ranks on one device), a missing library, KMCPG_RCCL=0 — the handle merges the shards' lists
// on the host instead (N device-to-host copies), which is what every multi-shard handle did before this file existed.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <stdlib.h>
#include <string.h>

#include <condition_variable>
#include <mutex>
#include <set>
#include <string>
#include <vector>

#include "engine.hpp"
#include "exchange.hpp"

namespace kmcpg {

namespace {
// the few RCCL entry points used (signatures of rccl.h, ROCm 7.2)
typedef struct ncclComm* ncclComm_t;
typedef int ncclResult_t;  // ncclSuccess == 0
constexpr int kNcclUint8 = 1;
struct Api {
  void* lib = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*Send)(const void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*Recv)(void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

bool load_api(Api* a, std::string* why) {
  static std::mutex mu;
  static Api cached;
  static bool tried = false, ok = false;
  static std::string err;
  std::lock_guard<std::mutex> g(mu);
  if (!tried) {
    tried = true;
    for (const char* name : {"librccl.so.1", "librccl.so"}) {
      cached.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
      if (cached.lib) break;
    }
    if (!cached.lib) err = std::string("dlopen(librccl.so.1): ") + (dlerror() ? dlerror() : "not found");
    else {
      auto sym = [&](const char* n) { return dlsym(cached.lib, n); };
      cached.CommInitAll = (decltype(cached.CommInitAll))sym("ncclCommInitAll");
      cached.CommDestroy = (decltype(cached.CommDestroy))sym("ncclCommDestroy");
      cached.Send = (decltype(cached.Send))sym("ncclSend");
      cached.Recv = (decltype(cached.Recv))sym("ncclRecv");
      cached.GroupStart = (decltype(cached.GroupStart))sym("ncclGroupStart");
      cached.GroupEnd = (decltype(cached.GroupEnd))sym("ncclGroupEnd");
      cached.GetErrorString = (decltype(cached.GetErrorString))sym("ncclGetErrorString");
      ok = cached.CommInitAll && cached.CommDestroy && cached.Send && cached.Recv && cached.GroupStart && cached.GroupEnd && cached.GetErrorString;
      if (!ok) err = "librccl.so.1 lacks an expected symbol";
    }
  }
  *a = cached;
  *why = err;
  return ok;
}
}  // namespace

// one gather in flight: its concatenation buffer on devices[0] and the events that say when its transfers are done
struct GatherSlot {
  DevBuf<uint8_t> buf;               // on devices[0]
  hipEvent_t done = nullptr;         // devices[0]: the receives (and what on_device queued behind them) have completed
  std::vector<hipEvent_t> sent;      // per device: its send has left the source buffer
  bool busy = false;
};

struct Exchange {
  Api api;
  std::vector<int> devices;
  std::vector<ncclComm_t> comms;
  std::vector<hipStream_t> streams;  // one per device: the exchange never queues behind the next batches' kernels
  std::mutex mu;                     // group calls on one set of communicators are serialised: held from ncclGroupStart to ncclGroupEnd only
  std::condition_variable cv;        // a slot came back
  GatherSlot slots[4];               // as many gathers in flight as a handle has lanes
  std::string note;
};

const char* exchange_note(const Exchange* x) { return x ? x->note.c_str() : "host merge"; }

// Brings parts[r] (count[r] bytes at device pointer src[r] on devices[r]) together on devices[0] and copies the concatenation to
// `dst` (pinned host memory, sum of the counts) — rank order.  Every src[r] must be complete (its kernels waited for).
// Several host threads may be in here at once (one per batch in flight): each takes a gather slot of its own; only the group call
// itself — ncclGroupStart ... ncclGroupEnd on the shared communicators — runs under the mutex.  What follows (on_device's kernels
// and copies, the waits) is ordered by the streams and by the slot's events, so one batch's K3 and D2H overlap the next batch's
// transfers instead of holding them up.
int exchange_gather(Exchange* x, const std::vector<const void*>& src, const std::vector<uint64_t>& bytes, uint8_t* dst, const ExchangeOnDevice& on_device) {
  const int n = (int)x->devices.size();
  if ((int)src.size() != n || (int)bytes.size() != n) return kmcpg_fail(KMCPG_EINVAL, "exchange: %zu parts for %d devices", src.size(), n);
  uint64_t total = 0;
  for (uint64_t b : bytes) total += b;
  if (total == 0) return 0;
  GatherSlot* sl = nullptr;
  {
    std::unique_lock<std::mutex> g(x->mu);
    x->cv.wait(g, [&] {
      for (auto& s : x->slots)
        if (!s.busy) {
          sl = &s;
          return true;
        }
      return false;
    });
    sl->busy = true;
  }
  struct Release {  // every way out hands the slot back
    Exchange* x;
    GatherSlot* sl;
    ~Release() {
      {
        std::lock_guard<std::mutex> g(x->mu);
        sl->busy = false;
      }
      x->cv.notify_one();
    }
  } release{x, sl};
  HIPCHK(hipSetDevice(x->devices[0]));
  if (sl->buf.ensure(total + 32)) return kmcpg_fail(KMCPG_ENOMEM, "hipMalloc failed");
  if (!sl->done) HIPCHK(hipEventCreateWithFlags(&sl->done, hipEventDisableTiming));
  if (sl->sent.empty()) {
    sl->sent.assign((size_t)n, nullptr);
    for (int r = 1; r < n; r++) {
      HIPCHK(hipSetDevice(x->devices[r]));
      HIPCHK(hipEventCreateWithFlags(&sl->sent[(size_t)r], hipEventDisableTiming));
    }
    HIPCHK(hipSetDevice(x->devices[0]));
  }
  uint8_t* const d_cat = sl->buf.p + 16;  // 16 bytes in front of the concatenation belong to on_device (a count word)
  auto chk = [&](ncclResult_t r, const char* what) { return r == 0 ? 0 : kmcpg_fail(KMCPG_EDEVICE, "RCCL %s: %s", what, x->api.GetErrorString(r)); };
  int rc = 0;
  {
    std::lock_guard<std::mutex> g(x->mu);
    if (int rc0 = chk(x->api.GroupStart(), "ncclGroupStart")) return rc0;
    uint64_t off = 0;
    for (int r = 0; r < n && rc == 0; r++) {
      if (bytes[r] == 0) continue;
      // both halves of every transfer inside one group: rank r sends its list, rank 0 receives it at its place in the concatenation
      // (r == 0 is a send to itself, which RCCL turns into a device-local copy)
      (void)hipSetDevice(x->devices[r]);
      rc = chk(x->api.Send(src[r], bytes[r], kNcclUint8, 0, x->comms[(size_t)r], x->streams[(size_t)r]), "ncclSend");
      if (rc) break;
      (void)hipSetDevice(x->devices[0]);
      rc = chk(x->api.Recv(d_cat + off, bytes[r], kNcclUint8, r, x->comms[0], x->streams[0]), "ncclRecv");
      off += bytes[r];
    }
    const int rc_end = chk(x->api.GroupEnd(), "ncclGroupEnd");
    if (rc == 0) rc = rc_end;
    // this gather's sends, marked on the senders' streams before anybody else's group can follow them
    for (int r = 1; r < n && rc == 0; r++) {
      if (hipSetDevice(x->devices[r]) != hipSuccess || hipEventRecord(sl->sent[(size_t)r], x->streams[(size_t)r]) != hipSuccess)
        rc = kmcpg_fail(KMCPG_EDEVICE, "exchange: hipEventRecord failed on device %d", x->devices[r]);
    }
  }
  if (rc) {  // something may be queued: nothing of this slot is handed back while it could still be written
    for (int r = 0; r < n; r++)
      if (hipSetDevice(x->devices[r]) == hipSuccess) (void)hipStreamSynchronize(x->streams[(size_t)r]);
    (void)hipSetDevice(x->devices[0]);
    return rc;
  }
  HIPCHK(hipSetDevice(x->devices[0]));
  if (on_device) {
    rc = on_device(d_cat, total, (void*)x->streams[0]);
  } else {
    const hipError_t e = hipMemcpyAsync(dst, d_cat, total, hipMemcpyDeviceToHost, x->streams[0]);
    if (e != hipSuccess) rc = kmcpg_fail(KMCPG_EDEVICE, "exchange: hipMemcpyAsync: %s", hipGetErrorString(e));
  }
  // (stream order: everything this call queued on streams[0] precedes the event, whatever other gathers queued in between)
  hipError_t e = hipEventRecord(sl->done, x->streams[0]);
  if (e == hipSuccess) e = hipEventSynchronize(sl->done);
  else (void)hipStreamSynchronize(x->streams[0]);
  for (int r = 1; r < n; r++) {  // the senders' buffers are free again once their sends have completed
    if (hipSetDevice(x->devices[r]) != hipSuccess || hipEventSynchronize(sl->sent[(size_t)r]) != hipSuccess) {
      if (e == hipSuccess) e = hipErrorUnknown;
    }
  }
  (void)hipSetDevice(x->devices[0]);
  if (rc) return rc;
  HIPCHK(e);
  return 0;
}

void exchange_destroy(Exchange* x) {
  if (!x) return;
  for (size_t r = 0; r < x->comms.size(); r++)
    if (x->comms[r]) {
      (void)hipSetDevice(x->devices[r]);
      (void)x->api.CommDestroy(x->comms[r]);
    }
  for (size_t r = 0; r < x->streams.size(); r++)
    if (x->streams[r]) {
      (void)hipSetDevice(x->devices[r]);
      (void)hipStreamDestroy(x->streams[r]);
    }
  for (auto& sl : x->slots) {
    for (size_t r = 1; r < sl.sent.size(); r++)
      if (sl.sent[r]) {
        (void)hipSetDevice(x->devices[r]);
        (void)hipEventDestroy(sl.sent[r]);
      }
    if (!x->devices.empty()) (void)hipSetDevice(x->devices[0]);
    if (sl.done) (void)hipEventDestroy(sl.done);
    sl.buf.release();
  }
  delete x;
}

// nullptr (and *why) when the handle should merge on the host.  KMCPG_RCCL: 0 = never, 1 (default) = when the devices are
// distinct and there are at least two, force = also for a single device (tests on a one-GPU box: send/recv to self).
Exchange* exchange_create(const std::vector<int>& devices, std::string* why) {
  const char* env = getenv("KMCPG_RCCL");
  const bool forced = env && strcmp(env, "force") == 0;
  if (env && !forced && atoi(env) == 0) {
    *why = "KMCPG_RCCL=0";
    return nullptr;
  }
  if (devices.size() < 2 && !forced) {
    *why = "one device";
    return nullptr;
  }
  if (std::set<int>(devices.begin(), devices.end()).size() != devices.size()) {
    *why = "duplicate device ordinals (RCCL wants one rank per device)";
    return nullptr;
  }
  Api api;
  if (!load_api(&api, why)) return nullptr;
  Exchange* x = new Exchange();
  x->api = api;
  x->devices = devices;
  x->comms.assign(devices.size(), nullptr);
  x->streams.assign(devices.size(), nullptr);
  ncclResult_t r = api.CommInitAll(x->comms.data(), (int)devices.size(), devices.data());
  if (r != 0) {
    *why = std::string("ncclCommInitAll: ") + api.GetErrorString(r);
    for (auto& c : x->comms) c = nullptr;
    exchange_destroy(x);
    return nullptr;
  }
  for (size_t i = 0; i < devices.size(); i++) {
    // highest priority: the few hundred KB of a batch's hit lists must not wait behind the next batch's chip-filling kernels
    int prio_lo = 0, prio_hi = 0;
    if (hipSetDevice(devices[i]) != hipSuccess || hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi) != hipSuccess ||
        hipStreamCreateWithPriority(&x->streams[i], hipStreamNonBlocking, prio_hi) != hipSuccess) {
      *why = "stream creation failed";
      exchange_destroy(x);
      return nullptr;
    }
  }
  // self-test: every rank sends 64 bytes that name it; rank 0 must see them in rank order
  {
    const int n = (int)devices.size();
    std::vector<void*> bufs((size_t)n, nullptr);
    std::vector<const void*> src;
    std::vector<uint64_t> bytes((size_t)n, 64);
    std::vector<uint8_t> pattern(64), got((size_t)n * 64, 0);
    bool ok = true;
    for (int i = 0; i < n && ok; i++) {
      memset(pattern.data(), 0xA0 + i, 64);
      ok = hipSetDevice(devices[(size_t)i]) == hipSuccess && hipMalloc(&bufs[(size_t)i], 64) == hipSuccess &&
           hipMemcpy(bufs[(size_t)i], pattern.data(), 64, hipMemcpyHostToDevice) == hipSuccess;
      src.push_back(bufs[(size_t)i]);
    }
    uint8_t* pinned = nullptr;
    ok = ok && hipHostMalloc((void**)&pinned, (size_t)n * 64, hipHostMallocDefault) == hipSuccess;
    if (ok) ok = exchange_gather(x, src, bytes, pinned) == 0;
    if (ok) {
      memcpy(got.data(), pinned, got.size());
      for (int i = 0; i < n; i++)
        for (int j = 0; j < 64; j++) ok = ok && got[(size_t)i * 64 + j] == (uint8_t)(0xA0 + i);
    }
    if (pinned) (void)hipHostFree(pinned);
    for (int i = 0; i < n; i++)
      if (bufs[(size_t)i]) {
        (void)hipSetDevice(devices[(size_t)i]);
        (void)hipFree(bufs[(size_t)i]);
      }
    if (!ok) {
      *why = "RCCL self-test failed: " + kmcpg_err_ref();
      exchange_destroy(x);
      return nullptr;
    }
  }
  x->note = "RCCL gather over " + std::to_string(devices.size()) + " device(s)";
  return x;
}

}  // namespace kmcpg
