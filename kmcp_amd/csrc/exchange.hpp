// exchange.hpp — RCCL hit-list gather of an in-process multi-GPU handle (exchange.cpp).
#pragma once
#include <stdint.h>

#include <string>
#include <vector>

namespace kmcpg {
struct Exchange;
Exchange* exchange_create(const std::vector<int>& devices, std::string* why);  // nullptr: merge on the host (*why says why)
void exchange_destroy(Exchange* x);
const char* exchange_note(const Exchange* x);
// `on_device` (optional): called instead of the copy to `dst`, with the concatenation still on devices[0] — 16-byte aligned, an
// 8-byte word in front of it (d_cat - 16) free for the caller — and the exchange stream of that device: what it enqueues there runs
// behind the receives; exchange_gather waits for what it queued before it returns (each gather in flight has a buffer of its own).
#include <functional>
typedef std::function<int(uint8_t* d_cat, uint64_t total_bytes, void* stream)> ExchangeOnDevice;
int exchange_gather(Exchange* x, const std::vector<const void*>& src, const std::vector<uint64_t>& bytes, uint8_t* dst, const ExchangeOnDevice& on_device = nullptr);
}  // namespace kmcpg
