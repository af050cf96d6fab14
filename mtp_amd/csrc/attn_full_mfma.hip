// placeholder TU until the full-attention MFMA kernels land (the f32-VALU kernels in attn.hip are used meanwhile)
#include "attn_mfma.h"
#include "common.h"

int mtp_full_fwd_mfma_launch(const void*, void*, float*, const float*, const float*, int64_t, int64_t, int64_t, int64_t, float, hipStream_t) { return MTP_ERR_UNSUPPORTED; }
int mtp_full_bwd_mfma_launch(const void*, const void*, const void*, const float*, void*, const float*, const float*, float*, int64_t, int64_t, int64_t, int64_t, float, hipStream_t) {
    return MTP_ERR_UNSUPPORTED;
}
