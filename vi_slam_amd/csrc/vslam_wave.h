/* vslam_wave.h -- wave64 / instruction-selection helpers shared by the kernels (device code only). */
#ifndef VSLAM_WAVE_H
#define VSLAM_WAVE_H
#include <hip/hip_runtime.h>
#include <stdint.h>

/* wave64 min-reduction on DPP (no LDS traffic, a few cycles per step instead of a ds_bpermute round trip):
 * quad swaps, row rotations, then row_bcast:15 / row_bcast:31; the result sits in lane 63 and is returned
 * wave-uniform through v_readlane. */
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ uint32_t dpp_mov(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, ROW_MASK, 0xf, false);
}
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
    v = min(v, dpp_mov<0xb1, 0xf>(v));  /* quad_perm [1,0,3,2] */
    v = min(v, dpp_mov<0x4e, 0xf>(v));  /* quad_perm [2,3,0,1] */
    v = min(v, dpp_mov<0x124, 0xf>(v)); /* row_ror:4 */
    v = min(v, dpp_mov<0x128, 0xf>(v)); /* row_ror:8 */
    v = min(v, dpp_mov<0x142, 0xa>(v)); /* row_bcast:15 into rows 1,3 */
    v = min(v, dpp_mov<0x143, 0xc>(v)); /* row_bcast:31 into rows 2,3 */
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}

/* vslam_tuning.wave_prio: the narrow kernels on a step's dependency chain (quadtree, output order, descriptors,
 * matchers) raise their wave priority at entry, so that next to the grid-filling FAST / blur / pyramid waves of the
 * other contexts -- which can issue at any time -- the SIMD's arbiter (priority, then age) hands them the issue slots.
 * `on` is a kernel argument: a scalar branch around one s_setprio. */
__device__ __forceinline__ void wave_prio_raise(int on) {
    if (on) __builtin_amdgcn_s_setprio(3);
}

/* a * b + c as ONE instruction on the 24-bit multiplier (hipcc turns __umul24 back into a 32-bit multiply plus adds, or a
 * 64-bit v_mad_u64_u32 chain for size_t address arithmetic, where it can prove the operands small; all of them issue at the
 * same rate on gfx950 -- profiles/r04_issue_rate_multiplies.txt -- so this is about instruction COUNT).  _s: b is wave-uniform
 * (an SGPR operand) */
__device__ __forceinline__ uint32_t mad24u_s(uint32_t a, uint32_t b_uniform, uint32_t c) {
    uint32_t r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(b_uniform), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t mad24u(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ int mad24i(int a, int b, int c) {
    int r;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

#endif /* VSLAM_WAVE_H */
