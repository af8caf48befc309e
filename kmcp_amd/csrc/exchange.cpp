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

struct Exchange {
  Api api;
  std::vector<int> devices;
  std::vector<ncclComm_t> comms;
  std::vector<hipStream_t> streams;  // one per device: the exchange never queues behind the next batches' kernels
  std::mutex mu;                     // group calls on one set of communicators are serialised
  DevBuf<uint8_t> d_gather;          // on devices[0]
  std::string note;
};

const char* exchange_note(const Exchange* x) { return x ? x->note.c_str() : "host merge"; }

// Brings parts[r] (count[r] bytes at device pointer src[r] on devices[r]) together on devices[0] and copies the concatenation to
// `dst` (pinned host memory, sum of the counts) — rank order.  Every src[r] must be complete (its kernels waited for).
int exchange_gather(Exchange* x, const std::vector<const void*>& src, const std::vector<uint64_t>& bytes, uint8_t* dst, const ExchangeOnDevice& on_device) {
  const int n = (int)x->devices.size();
  if ((int)src.size() != n || (int)bytes.size() != n) return kmcpg_fail(KMCPG_EINVAL, "exchange: %zu parts for %d devices", src.size(), n);
  uint64_t total = 0;
  for (uint64_t b : bytes) total += b;
  if (total == 0) return 0;
  std::lock_guard<std::mutex> g(x->mu);
  HIPCHK(hipSetDevice(x->devices[0]));
  if (x->d_gather.ensure(total + 32)) return kmcpg_fail(KMCPG_ENOMEM, "hipMalloc failed");
  uint8_t* const d_cat = x->d_gather.p + 16;  // 16 bytes in front of the concatenation belong to on_device (a count word)
  auto chk = [&](ncclResult_t r, const char* what) { return r == 0 ? 0 : kmcpg_fail(KMCPG_EDEVICE, "RCCL %s: %s", what, x->api.GetErrorString(r)); };
  if (int rc = chk(x->api.GroupStart(), "ncclGroupStart")) return rc;
  int rc = 0;
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
  if (rc) return rc;
  HIPCHK(hipSetDevice(x->devices[0]));
  if (on_device) {
    rc = on_device(d_cat, total, (void*)x->streams[0]);
    const hipError_t e = hipStreamSynchronize(x->streams[0]);
    if (rc) return rc;
    HIPCHK(e);
  } else {
    HIPCHK(hipMemcpyAsync(dst, d_cat, total, hipMemcpyDeviceToHost, x->streams[0]));
    HIPCHK(hipStreamSynchronize(x->streams[0]));
  }
  for (int r = 1; r < n; r++) {  // the senders' buffers are free again once their streams have drained
    HIPCHK(hipSetDevice(x->devices[r]));
    HIPCHK(hipStreamSynchronize(x->streams[(size_t)r]));
  }
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
  if (!x->devices.empty()) (void)hipSetDevice(x->devices[0]);
  x->d_gather.release();
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
