/* ia2p_debug.h -- test / tuning hooks and the per-kernel timing interface of libia2p_hip.so.
 *
 * NOT part of the drop-in boundary (include/ia2p.h): nothing here stands in for a reference interface. The hooks pin a tile variant / K-split / fusion decision
 * so that tests can prove every route gives the same bits (tests/test_ops_gpu.py, test_gn_fused_gpu.py), and the profile interface is how bench.py times
 * kernels live with HIP events on the launch stream for its `roofline` block. Same library, same conventions (ia2p.h head comment).
 */
#ifndef IA2P_DEBUG_H
#define IA2P_DEBUG_H

#include "ia2p.h"

#ifdef __cplusplus
extern "C" {
#endif

void ia2p_debug_set_gemm_splitk(int splitk);  /* -1 auto (tests / tuning; engine path only) */
/* K-split ticket counters (one buffer per device and stream) start a new epoch: each stream's buffer is re-zeroed, on that stream, in front of its next
 * K-split launch. The library does this itself when a context is created and whenever it reports IA2P_ERR_HIP (a launch that died mid-flight may have left
 * tickets behind); exported for tests and for hosts that catch a device error outside the library. */
void ia2p_debug_invalidate_splitk_counters(void);
int ia2p_debug_fill_splitk_counters(void* stream, int value);   /* tests: every ticket of the stream's buffer := value, on the stream (non-zero = what a dead launch leaves); 0 / -1 */
void ia2p_debug_set_splitk_inkernel(long long bytes); /* slab-set size (splitk*M*N*4) up to which a K split combines inside the GEMM launch; < 0: IA2P_SPLITK_INKERNEL / default (tests, A/B runs) */
void ia2p_debug_set_gn_plan(int mode);         /* GroupNorm fused into its convolution: -1 as the measured plan of the site says (default), 1 wherever the site's tile is a halo-staged one, 0 nowhere (tests) */
void ia2p_debug_set_gemm_tile(int variant);   /* -1 auto; else index into IA2P_GEMM_TILES of csrc/common.h, 0..26 (tests / tuning) */
/* fused to_q + cross-attention: contexts created AFTER this call fuse launches of at least `tiles` 128-query x head tiles (-1: the built-in 128). Tests only:
 * lets a tiny model take the fused path. */
void ia2p_debug_set_xattn_min_tiles(int tiles);
/* IP-Adapter cross-attention (reference attention_processor.py:371,387,397): 1 = the image-token keys ride in the free slots of the last text-key tile (81 keys = two tiles), 0 = a tile of
 * their own (three), -1 = IA2P_ATTN_FOLD or the default (1). Tests: the two forms agree to one fp16 ulp (the two softmaxes are the same numbers; sums of the probabilities differ in order). */
void ia2p_debug_set_attn_fold(int mode);
/* the tile table (tests / tools): out[4] = {tile rows, tile columns, LDS ring stages, schedule: 0 plain, 1 ping-pong, 2 eight-phase, 3 halo-staged 3x3 convolution (ping-pong over 16 x 16 pixel patches)}; 0, or -1 past the last variant */
int ia2p_debug_gemm_tile_info(int variant, int* out);
/* the tile variant and K-split the library picks for a problem (pure function of the shape; host-only, no GPU needed) */
void ia2p_debug_gemm_plan(int M, int N, int K, int conv, int geglu, int* variant, int* splitk);

/* ---- per-kernel timing (bench.py roofline leg): HIP events on the launch stream around each launch ------------------
 * Classes are device kernel names as rocprofv3 prints them (e.g. "gemm_f16_kernel<128, 64, false>"). */
ia2p_status ia2p_profile_enable(ia2p_ctx* ctx, int on);   /* also clears the sums */
int ia2p_profile_classes(void);
/* sums since enable for class k: launches, milliseconds, algorithmic flops and bytes */
/* bytes of the NEXT contraction's weights that the launches of class k streamed with their trailing prefetch workgroups (counted in the class's
 * HBM-side traffic by the PMC counters, but not its own operands) */
ia2p_status ia2p_profile_read_prefetch(ia2p_ctx* ctx, int k, double* bytes);
ia2p_status ia2p_profile_read(ia2p_ctx* ctx, int k, char* name, int name_len, int64_t* launches, double* ms, double* flops, double* bytes);
/* the same sums by REGION of the UNet evaluation: 0 = other (embeddings), 1 = conv blocks (conv_in / conv_out, the ResnetBlock2Ds with their
 * GroupNorm+SiLU and 1x1 shortcuts, resample convolutions, skip concatenation -- SURVEY.md §8d "conv blocks"), 2 = transformer blocks */
ia2p_status ia2p_profile_read_region(ia2p_ctx* ctx, int region, int64_t* launches, double* ms, double* flops, double* bytes);
/* the same sums by layer ROLE, i.e. by the executor's call site, whatever tile / fusion the plan table picked for the launch: 0 other, 1 FF-in (GEGLU projection),
 * 2 FF-out, 3 QKV + self-attention, 4 attention out-projections (reference attention_processor.py:267,400), 5 to_q + cross-attention (:344,371,387,397),
 * 6 3x3 convolutions (ResnetBlock2D convs incl. the fused shortcut, resample convs), 7 GroupNorm(+SiLU), 8 proj_in / proj_out, 9 context K/V projection (:358-359,379-380),
 * 10 time / add embeddings, 11 conv_in / conv_out. bench.py keys `roofline` by role: the dominant kernel INSTANTIATION flips with the tuner's picks, the role does not. */
int ia2p_profile_roles(void);
ia2p_status ia2p_profile_read_role(ia2p_ctx* ctx, int role, char* name, int name_len, int64_t* launches, double* ms, double* flops, double* bytes);
/* launches / milliseconds of role `role` that ran on kernel class k (ia2p_profile_read's index): which instantiations carried the role on this plan table */
ia2p_status ia2p_profile_read_role_class(ia2p_ctx* ctx, int role, int k, int64_t* launches, double* ms);

#ifdef __cplusplus
}
#endif
#endif /* IA2P_DEBUG_H */
