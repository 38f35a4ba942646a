// MFMA (fp32 matrix-core) fast path of the grouped decoder -- placeholder until the
// tuned kernel lands; the generic path in generic.hip serves every shape meanwhile.
#include "common.h"

namespace oivae {

bool mfma_decoder_applies(const oivae_shape_t&) { return false; }

int launch_decoder_mfma(const oivae_shape_t&, const oivae_tensors_t*, const oivae_tensors_t*, const float*,
                        int64_t, int32_t, const float*, float, float*, Workspace&, hipStream_t) {
  return fail(OIVAE_E_SHAPE, "MFMA decoder path not built");
}

}  // namespace oivae
