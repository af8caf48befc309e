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
int exchange_gather(Exchange* x, const std::vector<const void*>& src, const std::vector<uint64_t>& bytes, uint8_t* dst);
}  // namespace kmcpg
