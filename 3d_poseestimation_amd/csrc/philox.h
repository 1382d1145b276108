// Philox4x32-10 (Salmon et al., SC'11) -- the dropout stream of the HIP path.
//
// The reference draws dropout masks with torch.nn.Dropout (phase1_lifting/baselineModel.py:
// 20-21,37,43,81,94); torch's CPU Bernoulli stream cannot be reproduced on a GPU, so the
// path defines its own counter-based stream.  oracle/philox.py is its bit-exact CPU twin.
//
//   element (row r, col c) of hidden layer `layer`, H columns:
//     e = r*H + c;  g = e >> 2;  j = e & 3
//     counter = (g & 0xffffffff, g >> 32, layer, step & 0xffffffff)
//     key     = (seed & 0xffffffff, (seed >> 32) ^ (step >> 32))
//     keep    = philox4x32_10(counter, key)[j] >= thr,  thr = min(2^32-1, floor(p * 2^32))
//
// One call yields the decisions of four consecutive columns = one float4 of activations;
// nothing is stored: the backward pass reads the keep&relu bitmap the forward wrote.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pl {

struct Philox4 { uint32_t v[4]; };

__device__ __forceinline__ Philox4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                 uint32_t k0, uint32_t k1) {
  constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
    const uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0;
    const uint32_t n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += W0; k1 += W1;
  }
  Philox4 o;
  o.v[0] = c0; o.v[1] = c1; o.v[2] = c2; o.v[3] = c3;
  return o;
}

inline uint32_t dropout_threshold(float p) {
  double t = (double)p * 4294967296.0;
  if (t > 4294967295.0) t = 4294967295.0;
  if (t < 0.0) t = 0.0;
  return (uint32_t)t;   // floor
}

}  // namespace pl
