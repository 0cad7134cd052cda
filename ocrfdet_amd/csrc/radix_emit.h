// What the LAST pass of a radix sort (csrc/index_prep.hip) may write besides — or instead of — the sorted (key, value)
// pairs: the consumer's own arrays, so that no further launch has to turn the pairs into them.
#pragma once
#include <hip/hip_runtime.h>

namespace ocrf {

// render-plan build (csrc/raster_plan.hip): value = the record's Gaussian-major index e.  The per-view lists in blend
// order leave the sort as s_e = e, s_id = rec_id[e], s_key = depth bits of the key, s_pix = pixel centre of e.
struct RadixPlanEmit {
  const int* rec_id;
  const float4* e_q1;
  unsigned* s_e;
  unsigned* s_id;
  unsigned* s_key;
  float2* s_pix;
  unsigned depth_mask, key_base;
};

}  // namespace ocrf
