// device_utils.hpp — small device helpers shared by the kernel files.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "fastmod.hpp"
#include "nthash.hpp"

namespace kmcpg {

__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

}  // namespace kmcpg
