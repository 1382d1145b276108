// Pieces of the BatchNorm statistics finalize shared by the kernels that do it (elementwise.hip, small_layer.hip).
// Floating-point contraction is OFF inside them: "SyncBN == one process on the concatenated batch, bit for bit" is a tested
// property, and it must not depend on what the compiler contracts.
#pragma once
#include <hip/hip_runtime.h>

namespace pl {

__device__ __forceinline__ float bn_shift_of(float beta, float mean, float sc) {
#pragma clang fp contract(off)
  return beta - mean * sc;
}
// running statistics of nn.BatchNorm1d in training mode: momentum update with the UNBIASED batch variance
__device__ __forceinline__ void bn_running_update(float& rm, float& rv, float mean, float var, float Bt, float mo) {
#pragma clang fp contract(off)
  const float unbiased = var * (Bt / (Bt - 1.0f));
  rm = (1.0f - mo) * rm + mo * mean;
  rv = (1.0f - mo) * rv + mo * unbiased;
}

}  // namespace pl
