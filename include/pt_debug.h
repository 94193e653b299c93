/* include/pt_debug.h — developer surface of libpt_hip.so: tuning knobs, per-kernel timers, parity probes.
 *
 * Nothing here replaces a reference interface: the boundary a maintainer binds is include/pt_api.h.  These entry points exist for
 * bench.py (roofline timing), scripts/ (A/B runs) and tests/ (numeric-contract probes).
 */
#ifndef PT_DEBUG_H
#define PT_DEBUG_H
#include "pt_api.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Tuning knobs: 0 = path slots in flight (default 0 = automatic: one per job of a synchronous batch within [2^20, 2^22]; 5/8 of the
 * backlog up to 2^23 for overlapped batches), 1 = count traversal statistics (0/1), 2 = LDS bytes per block of the simple intersect
 * kernel, 3 = lanes of a wave waiting for their next BVH / retirement that make that phase worth a trip (default 8, 2 in the fused loop of the hand-written kernel; 1 = at once),
 * 4 = intersect kernel (0 simple, one block per 256 rays; 1 persistent blocks, compiled; 2 = default: the hand-written form of 1, csrc/hip/pt_extend_gfx950.s,
 *     for the launches it takes — at most 1024 BVHs, no empty leaves, ordered boxes, fewer than 2^23 - 1 inner nodes / triangle records, statistics off — and 1 for the others),
 * 5 = persistent block size (64/128/256/512/1024, default 256), 6 = persistent LDS tile bytes, 7 = idle lanes per wave that trigger a ray refill,
 * 8 = cap on the blocks per CU of the persistent grid (default 0 = no cap: as many as are resident at once, 8 blocks of 256 threads = 8 waves per SIMD),
 * 9 = the inner-node phase repeats while more than this many eighths of its starting lanes still sit on inner nodes (default 6),
 * 10 = inner-node records kept in breadth-first order (whole levels, the top of the trees); deeper ones are laid out depth-first,
 * 11 = width of the traversal-stack entries at least: 0 16-bit, 1 16 bits in LDS + 2 bits in registers (trees up to 131071 nodes), 2 the widest (compiled kernel: 32-bit;
 *      hand-written kernel: 16 bits + a byte in a second LDS array, trees up to 2^23 - 2 inner nodes / triangle records); -1 = automatic.
 * 14 = main loop of the hand-written intersect kernel: -1 automatic (default), 0 phase-voting like the compiled kernel, 1 fused trip (every lane on a node or
 *      a leaf advances each trip; a lane's record is requested the moment its entry is decided).
 * 17 = block size of the hand-written intersect kernel: 0 automatic (default: 1024 threads — 2 blocks per CU over a 32 KB tile of the trees' top — when the
 *      context has its GPU to itself and the launch gives every CU its two blocks, else 256 threads), 256, 1024.
 * 16 = numeric contract: 0 (default) exact — every float operation pinned, framebuffers bit-identical to the oracle; 1 relaxed — RNG, draw counts and
 *      branches as written, the continuous functions (Box-Muller's log / cos / sqrt, normalize, 1/d, 1/det, the BSDF weights' divides) on the
 *      hardware's v_log / v_cos / v_sqrt / v_rsq / v_rcp units: results within north_star's per-pixel RMSE <= 1e-3 of the oracle, not bit-identical.
 *      Applies to path tracing (RAYTRACING == 1) without statistics; takes effect with the next frame stream.
 * 18 = index-stack encoding of the path state (tests): 1 forces 8-bit dictionary codes (a 16-B state group) where the scene's refraction indices would
 *      fit the 3-bit codes that ride beside the throughput, 2 the ten floats themselves (three groups; what a scene with more than 254 distinct Ni gets);
 *      0 (default) automatic.
 * 19 = node records of the hand-written intersect kernel: -1 automatic (default: 80-B records whose plane pairs are stored in direction-sign order while the
 *      inner nodes fit the caches, 64-B records with the min/max step for larger trees), 0 80-B, 1 64-B.
 * 20 = per-ray cull of the object loop in the hand-written intersect kernel for scenes with more than 8 BVHs (one pass over at most 64 group boxes when a ray
 *      starts; the objects of a group whose box the ray misses are never tested): 1 (default) on, 0 off (every group box infinite: each root box is tested in turn).
 * Queries (tests): 12 = PT_OK iff the current scene runs on the hand-written intersect kernel (else PT_ERR_UNSUPPORTED and the reason in pt_last_error),
 * 13 = PT_OK iff that kernel has been launched more than `value` times by this context.
 * 21 = spatial partition, an experiment that is NOT the default (profiles/r06_c_cu_partition.txt: every split 25-68 % slower): e = 1..7 puts the intersect launches on a
 *      stream whose CU mask holds e eighths of every XCD's CUs and the shading launches on the complement (hipExtStreamCreateWithCUMask); 0 = off.
 */
int pt_set_option(pt_ctx* ctx, int option, int64_t value);

/* Per-kernel device time since pt_set_timing(1) / pt_reset_counters, measured with HIP events on the
 * launch stream: kernel 0 = intersect (extend), 1 = shade, 2 = pool start (revive), 3 = accumulate.
 * Synchronises.  launches = number of launches, total_ms = summed duration. */
int pt_kernel_time(pt_ctx* ctx, int kernel, int64_t* launches, double* total_ms);
int pt_set_timing(pt_ctx* ctx, int enabled);
/* Developer builds only (-DPT_PHASE_STATS, or -DPT_WAVE_STAMPS for the stamps alone; zeros otherwise): {trips, active lanes} of the persistent intersect kernel's phases
 * refill, next-object/retire, inner-node step, leaf step, and of its outer loop, since the last pt_reset_counters; from out[16] on
 * the 100 MHz finish and start times of the (up to 8192) waves of the last intersect launch.  Does not complete submitted batches. */
int pt_debug_phase_stats(pt_ctx* ctx, uint64_t* out, int n);
/* median launch duration of that kernel (steady-state figure: the mean also averages the short launches of a batch's tail) */
int pt_kernel_time_median(pt_ctx* ctx, int kernel, double* median_ms);

/* Debug / parity probes (used by tests): evaluates the device numeric contract.
 * fn: 0 sin, 1 cos, 2 log, 3 exp, 4 atan(x,y), 5 asin; 6 / 7 / 8 = the state after / the result of / random() of ONE NextRandom call
 * (frag.glsl:686-694), the uint32 state passed as float bits; 9 = byte / 255.0f as the texture samplers evaluate it (x = the byte as a float);
 * host pointers, n elements. */
int pt_debug_math(pt_ctx* ctx, int fn, const float* x, const float* y, float* out, size_t n);
/* Single rays through the intersect kernel option 4 selects (the production kernels included): o,d are n*3 f32 (host); out is n*4 f32 (t,u,v) + prim as int bits */
int pt_debug_intersect(pt_ctx* ctx, const float* o, const float* d, float* out, size_t n);

#ifdef __cplusplus
}
#endif
#endif
