// internal launchers of the bf16 MFMA attention kernels (attn_mfma.hip), called from the C-ABI entry points in attn.hip
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

int mtp_rvsa_fwd_mfma_launch(const void* qkv, const float* samp, void* o, float* lse, const float* rel_h, const float* rel_w, const float* bias_table,
                             int64_t B, int64_t Hp, int64_t Wp, int64_t heads, float scale, hipStream_t s);
int mtp_rvsa_bwd_mfma_launch(const void* qkv, const float* samp, const void* o, const void* dout, const float* lse, void* dqkv, float* dkv, float* dsamp,
                             float* rel_part, float* tab_part, const float* rel_h, const float* rel_w, const float* bias_table,
                             int64_t B, int64_t Hp, int64_t Wp, int64_t heads, float scale, hipStream_t s);
// how the 4-wave RVSA backward scatters dK_sel / dV_sel for this grid (attn_rvsa_bwd4.hip): 4 = separate dense-product kernel (no f32
// scratch: the caller skips its clearing / conversion passes), 1 / 0 = f32 atomics into the scratch, 2 = none
int mtp_rvsa_bwd_mfma_scatter_mode(int64_t Hp, int64_t Wp, int64_t heads);
int mtp_full_fwd_mfma_launch(const void* qkv, void* o, float* lse, const float* rel_h, const float* rel_w,
                             int64_t B, int64_t Hp, int64_t Wp, int64_t heads, float scale, hipStream_t s);
int mtp_full_bwd_mfma_launch(const void* qkv, const void* o, const void* dout, const float* lse, void* dqkv, const float* rel_h, const float* rel_w,
                             float* drel_part, int64_t B, int64_t Hp, int64_t Wp, int64_t heads, float scale, hipStream_t s);
// token grids of at most 16 x 16 (attn_full_v3.hip: row-aligned tiles, relative-position logits as MFMA k-slots)
bool mtp_full_v3_fits(int64_t Hp, int64_t Wp);
int mtp_full_v3_fwd_launch(const void* qkv, void* o, float* lse, const float* rel_h, const float* rel_w, int64_t B, int64_t Hp, int64_t Wp, int64_t heads,
                           float scale, hipStream_t s);
int mtp_full_v3_bwd_launch(const void* qkv, const void* o, const void* dout, const float* lse, void* dqkv, const float* rel_h, const float* rel_w,
                           float* drel_part, int64_t B, int64_t Hp, int64_t Wp, int64_t heads, float scale, hipStream_t s);
// beyond 256 tokens (attn_full_flash_bwd.hip); workspace as mtp_full_attn_bwd_workspace_floats
int mtp_full_bwd_flash_launch(const void* qkv, const void* o, const void* dout, const float* lse, void* dqkv, const float* rel_h, const float* rel_w,
                              float* drel_part, float* workspace, int64_t B, int64_t Hp, int64_t Wp, int64_t heads, float scale, hipStream_t s);

// bf16 I/O always takes the MFMA kernels; the f32-math VALU kernels of attn.hip serve f32 I/O (parity mode) and the grids the MFMA kernels
// do not take
static inline bool mtp_use_mfma_attn() { return true; }
