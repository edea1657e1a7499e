/* Entry points that exist only in the EXPERIMENTS build of libdiart_amd (hipcc -DDZ_EXPERIMENTS;
 * `python -m diart_amd.build --experiments` -> diart_amd/libdiart_amd_exp.so, loaded when DZ_EXPERIMENTS=1).
 * Not part of the drop-in boundary: these are the never-default kernel generations kept for measurements
 * (tools/g2bench.py, tools/g2ablate.py, tools/rec_contention.py; DESIGN.md "What was measured and left out").
 * In that build DZ_* environment variables select alternative kernels, some of them timing-only
 * instantiations whose RESULTS ARE WRONG (DZ_GP_DBG, DZ_CONV0_DBG, row_fragments >= 16 below).          */
#ifndef DIART_AMD_EXPERIMENTS_H
#define DIART_AMD_EXPERIMENTS_H
#include "diart_amd.h"
#ifdef __cplusplus
extern "C" {
#endif
/* generation 2 of the same layer (k_gemm_g2.hip: one accumulator per fragment, three LDS stages, counted
 * vmcnt); row_fragments = 2, 3, 4 -> 128 / 192 / 256 x 128 tiles, 0 = default.  dz_k_gemm_pre dispatches to it
 * with DZ_GEMM_GEN=2 outside the single-chunk latency regime.                                        */
int dz_k_gemm_g2(dz_ctx* ctx, const dz_convgemm_desc* desc, int row_fragments, void* stream);
/* generation 3 (k_gemm_g3.hip): the same loop as a persistent kernel, one workgroup per CU, every workgroup the
 * same number of k-tile iterations (a tile shared by two workgroups is finished by the one that holds its end);
 * row_fragments as above, 0 = default (4).  DZ_GEMM_GEN=3.  A timed-out hand-over is reported by dz_range_check
 * (error 7).                                                                                          */
int dz_k_gemm_g3(dz_ctx* ctx, const dz_convgemm_desc* desc, int row_fragments, void* stream);
/* measurement hook (tools/conv_pool_phases.py, tools/conv0_phases.py): while d_stamps != NULL, the dz_k_conv_pool and
 * dz_k_sinc_conv0_split launches record per-wave shader-clock stamps of a tile's phases (+ kernel entry / exit and the
 * wave's HW_ID) there: 64 x 8 bytes per stamped wave — conv_pool_h 512 x 2 waves, conv_pool_v2 256 x 8, sinc_conv0_v2
 * 512 x 4, the three-wave conv0 kernel 512 x 3                                                                        */
int dz_k_conv_pool_debug(long long* d_stamps);
/* The first SincNet stage (InstanceNorm -> 80 sinc filters, stride 10 -> |.| -> MaxPool(3)) of BOTH networks in one
 * launch, default precision only: the two models read the same window (the reference runs SincNet once per model,
 * /root/reference/src/diart/models.py:133,262) and their InstanceNorm1d(1) differ by the affine pair only, which
 * moves into the epilogue.  d_pair_planes: f16 planes [2][192][256] of the pair bank, d_pair_bsum[192] =
 * beta_net * sum of the slot's taps (diart_amd/weights.py pack_conv0_pair); d_moments from dz_wave_stats.  The next
 * dz_seg_forward* / dz_emb_frames of each handle (same batch, ordered behind this launch) starts at conv1.     */
int dz_sinc_conv0_pair(dz_seg* seg, dz_emb* emb, const float* d_wave, long long wave_stride, int batch,
                       const float* d_moments, const void* d_pair_planes, const float* d_pair_bsum, void* stream);
/* sinc_conv0 of both networks in one launch (kernel level; outputs as two dz_k_sinc_conv0_split calls)  */
int dz_k_sinc_conv0_pair(dz_ctx* ctx, const float* d_wave, long long stride, int batch, int samples,
                         const float* d_moments, const void* d_pair_planes, const float* d_pair_bsum, float gamma_seg,
                         float gamma_emb, float* d_y0_seg, float* d_y0_emb, float* d_part_seg, float* d_part_emb,
                         void* stream);
#ifdef __cplusplus
}
#endif
#endif
