// cmdgen_split.h - fp32-accurate tile GEMM on the bf16 matrix pipe of gfx950 ("split" engine).
//
// v_mfma_f32_32x32x2_f32 runs at the fp32 VECTOR rate (64 FLOP/clk/SIMD) and does not co-execute with
// VALU work; v_mfma_f32_32x32x16_bf16 runs at 16x that rate and holds the vector issue port for only 8 of
// its 32 cycles.  An fp32 value is the exact sum of three bf16 values (8 + 8 + 8 significant bits, each
// piece rounded to nearest-even from the remainder of the previous one):  a = a0 + a1 + a2, |a2| <= 2^-16 |a|.
// A product of two such values keeps every term down to 2^-16 |a||b| with SIX bf16 products
//     a b  ~=  a0 b0 + (a0 b1 + a1 b0) + (a0 b2 + a1 b1 + a2 b0)            (dropped: <= 3 * 2^-24 |a||b|)
// each of which is EXACT in the fp32 accumulator (8 x 8 bits), so the only rounding left is the fp32
// accumulation itself - the same kind and size of error an fp32 fmaf chain makes (measured against fp64 on
// [.,256] x [256,256] products: max 1.96e-6 / rms 1.87e-7 against 2.42e-6 / 2.24e-7 for the fmaf chain;
// tools/split_gemm_test.cpp, profiles/r02_m_split_gemm.txt).  Six bf16 MFMAs per 16 k-values cost 192
// cycles against 512 for the eight fp32 MFMAs they replace; delivered on MI355X (the chip lowers its clock
// under dense bf16 MFMA): 160-200 TF/s fp32-equivalent for a chain of [64,256] x [256,256] tile products
// against 157 TF/s PEAK for the fp32 instruction.
//
// Non-finite operands: NaN stays NaN; an INFINITE operand also yields NaN here (its second piece is Inf - Inf) where the
// fp32 instruction yields +-Inf - only reachable after an overflow, where the evaluation's NaN guard (dynamics.py:129-131)
// then resets the step instead of propagating an infinite velocity.
//
// Layouts
//   A (LDS): the fp32 tile exactly as the fp32-MFMA kernels keep it (rows x (K + 4) floats).  Every wave
//            splits the fragments it reads in registers (5.5 VALU operations per element, issued in the
//            shadow of the MFMAs), so producers and epilogues of the tile kernels do not change and a
//            64-row tile still needs 66 KB (two workgroups per CU).
//   B (global, L2-resident): per Linear weight W[out][in]
//            Ws[((nt * KB16 + kb) * 3 + s) * 64 + lane] = 8 bf16 { W_s[o][k .. k+7] },
//            o = 32 nt + (lane & 31), k = 16 kb + 8 (lane >> 5): one 16-byte load per lane and piece,
//            the three pieces of a fragment contiguous (3 KiB per (nt, kb)); split once on the host.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __attribute__((ext_vector_type(8))) __bf16 sbf16x8;
typedef __attribute__((ext_vector_type(2))) __bf16 sbf16x2;
typedef float sf32x16 __attribute__((ext_vector_type(16)));

// two floats -> packed bf16 pair (round to nearest even; v_cvt_pk_bf16_f32)
__device__ __forceinline__ uint32_t cvt_pk_bf16(float a, float b) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 v = {a, b};
    const sbf16x2 h = __builtin_convertvector(v, sbf16x2);
    return __builtin_bit_cast(uint32_t, h);
}
// the three bf16 pieces of two floats, each piece packed {a, b}
__device__ __forceinline__ void split3_pair(float a, float b, uint32_t& p0, uint32_t& p1, uint32_t& p2) {
    p0 = cvt_pk_bf16(a, b);
    const float ra = a - __uint_as_float(p0 << 16), rb = b - __uint_as_float(p0 & 0xffff0000u);     // exact
    p1 = cvt_pk_bf16(ra, rb);
    const float qa = ra - __uint_as_float(p1 << 16), qb = rb - __uint_as_float(p1 & 0xffff0000u);   // exact
    p2 = cvt_pk_bf16(qa, qb);
}
// ---------------------------------------------------------------------------------------------
// "Half" engine (round 5): TWO fp16 pieces per operand, THREE v_mfma_f32_32x32x16_f16 per fp32 product.
//   a = a0 + a1, a0 = fp16(a), a1 = fp16(a - a0):   |a - a0 - a1| <= 2^-22 |a|   (11 + 11 significant bits)
//   a b ~= a1 b0 + a0 b1 + a0 b0                    (dropped: a1 b1 <= 2^-22 |a||b|; every kept product is exact in the fp32 accumulator)
// Half the matrix instructions of the three-piece bf16 split and 2 instead of 5.5 vector operations per split element.  What it gives
// up is two bits of operand precision - below the rounding of the fp32 accumulation both engines and the reference share: emulated on
// [2048, 256] x [256, 256] products of SiLU activations (tools/half_split_error.py) the result is 7.8e-8 rms from the exact product
// against 9.4e-8 for the six-product bf16 split (twice as many accumulator roundings) and 1.6e-7 for an fp32 fmaf chain / BLAS sgemm.
// Range: fp16 ends at 65504.  Weights are pre-multiplied by a power of two (WPack::wh_scale: the largest weight in [2^11, 2^12)) so that
// both pieces are normal numbers, and the epilogue folds 1 / wh_scale into constants it multiplies by anyway.  An ACTIVATION above 65504
// becomes Inf and its second piece NaN: the evaluation's NaN guard then resets the step (dynamics.py:129-131), as for an infinite
// activation on the bf16 split; no fixture or chain of this repository comes within three orders of magnitude of that.
// ---------------------------------------------------------------------------------------------
typedef _Float16 sf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 sf16x2 __attribute__((ext_vector_type(2)));
// two floats -> packed fp16 pair (round to nearest even; v_cvt_pk_f16_f32)
__device__ __forceinline__ uint32_t cvt_pk_f16(float a, float b) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2 v = {a, b};
    const sf16x2 h = __builtin_convertvector(v, sf16x2);
    return __builtin_bit_cast(uint32_t, h);
}
// a - (float)half, the half taken from the low / high 16 bits of p: one v_fma_mix_f32 each (exact: the difference is representable)
__device__ __forceinline__ float sub_half_lo(float a, uint32_t p) {
    float r; asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(p), "v"(a)); return r;
}
__device__ __forceinline__ float sub_half_hi(float a, uint32_t p) {
    float r; asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r) : "v"(p), "v"(a)); return r;
}
// the two fp16 pieces of two floats, each piece packed {a, b}: 1.5 vector operations per element (round 6: the residual a - (float)a0 is
// formed AND rounded to fp16 by one v_fma_mixlo_f16 / v_fma_mixhi_f16 - the difference is exact in fp32, so the single rounding gives the
// bits of the former v_fma_mix_f32 + v_cvt_pk_f16_f32 pair)
__device__ __forceinline__ void split2_pair(float a, float b, uint32_t& p0, uint32_t& p1) {
    p0 = cvt_pk_f16(a, b);
    uint32_t r;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r) : "v"(p0), "v"(a));
    asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(r) : "v"(p0), "v"(b));
    p1 = r;
}
// four consecutive k-values of one row -> the two fp16 planes (8 bytes each)
__device__ __forceinline__ void split_store4_half(unsigned short* planes, int plane_elems, int off, const float4& v) {
    uint32_t a0, a1, b0, b1;
    split2_pair(v.x, v.y, a0, a1);
    split2_pair(v.z, v.w, b0, b1);
    *reinterpret_cast<uint2*>(planes + off) = make_uint2(a0, b0);
    *reinterpret_cast<uint2*>(planes + plane_elems + off) = make_uint2(a1, b1);
}

struct SFragPtr { const sbf16x8* p; unsigned ns; };     // a wave's first n-tile; ns = stride between n-tiles (16-byte units)

// cg = the wave's 64-column group; kb16_total = K / 16 of the packed matrix; kb0 = first k-block of this GEMM
__device__ __forceinline__ SFragPtr sfrag_ptr(const void* Ws, int kb16_total, int kb0, int cg) {
    const int lane = threadIdx.x & 63;
    SFragPtr f;
    f.p = reinterpret_cast<const sbf16x8*>(Ws) + ((size_t)(2 * cg) * kb16_total + kb0) * 192 + lane;
    f.ns = (unsigned)kb16_total * 192u;
    return f;
}

// B fragments in flight: two register sets of [2 n-tiles][3 pieces].  A GEMM enters with its k-block 0 in set 0
// (fetched by the previous GEMM's last block or by split_prefetch) and leaves with block 0 of `next` there.
struct SCarry { sbf16x8 b[2][2][3]; };

template <int NPC = 3>
__device__ __forceinline__ void split_load_set(const sbf16x8* q0, const sbf16x8* q1, sbf16x8 (&dst)[2][3]) {
#pragma unroll
    for (int s = 0; s < NPC; ++s) { dst[0][s] = q0[s * 64]; dst[1][s] = q1[s * 64]; }
}
template <int NPC = 3>
__device__ __forceinline__ void split_prefetch(const SFragPtr& f, SCarry& c) { split_load_set<NPC>(f.p, f.p + f.ns, c.b[0]); }

// eight consecutive k-values of one row (two float4) -> the three bf16 fragments of v_mfma_f32_32x32x16_bf16
__device__ __forceinline__ void split8(const float4& lo, const float4& hi, sbf16x8& p0, sbf16x8& p1, sbf16x8& p2) {
    uint32_t a[4], b[4], c[4];
    split3_pair(lo.x, lo.y, a[0], b[0], c[0]);
    split3_pair(lo.z, lo.w, a[1], b[1], c[1]);
    split3_pair(hi.x, hi.y, a[2], b[2], c[2]);
    split3_pair(hi.z, hi.w, a[3], b[3], c[3]);
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const u4 va = {a[0], a[1], a[2], a[3]}, vb = {b[0], b[1], b[2], b[3]}, vc = {c[0], c[1], c[2], c[3]};
    p0 = __builtin_bit_cast(sbf16x8, va); p1 = __builtin_bit_cast(sbf16x8, vb); p2 = __builtin_bit_cast(sbf16x8, vc);
}

// acc[m][n] += A(lds fp32, row stride lda floats) x W^T over KB16 * 16 k-values; MT = 64 or 32 rows, the wave's 64 columns.
// The k loop is rolled (two k-blocks per iteration, running pointers): fully unrolled, every fragment address of a GEMM
// becomes a loop-invariant 64-bit value that the persistent tile loops hoist and spill, and seven unrolled GEMMs of the
// node kernel would not fit the instruction cache.  Per block: the weight fragments of the NEXT block are requested
// (one block = 24 MFMAs = 768 cycles at 64 rows ahead), the fp32 A fragments of the block after next are read from
// LDS, and the split of the next block's A fragments is interleaved with this block's MFMAs.
template <int MT, int KB16>
__device__ __forceinline__ void tile_gemm_rsplit(const float* ldsA, int lda, const SFragPtr cur, const SFragPtr next,
                                                 sf32x16 (&acc)[MT / 32][2], SCarry& carry) {
    static_assert(KB16 % 2 == 0, "K must be a multiple of 32");
    constexpr int NMT = MT / 32;
    const int lane = threadIdx.x & 63;
    const float* ap = ldsA + (lane & 31) * lda + (lane >> 5) * 8;
    const sbf16x8* q0 = cur.p + 192;                  // k-block 1
    const sbf16x8* q1 = q0 + cur.ns;
    float4 raw[2][NMT][2];                            // fp32 fragments of the next two k-blocks
    sbf16x8 a[2][NMT][3];                             // split fragments of this and the next k-block
#define RS_LOADA(SET, PTR)                                                                                           \
    _Pragma("unroll") for (int m = 0; m < NMT; ++m) {                                                                \
        raw[SET][m][0] = *reinterpret_cast<const float4*>((PTR) + m * 32 * lda);                                     \
        raw[SET][m][1] = *reinterpret_cast<const float4*>((PTR) + m * 32 * lda + 4); }
#define RS_SPLIT(DST, SET)                                                                                           \
    _Pragma("unroll") for (int m = 0; m < NMT; ++m) split8(raw[SET][m][0], raw[SET][m][1], a[DST][m][0], a[DST][m][1], a[DST][m][2]);
#define RS_MFMAS(AS, BS)                                                                                             \
    _Pragma("unroll") for (int m = 0; m < NMT; ++m)                                                                  \
        _Pragma("unroll") for (int n = 0; n < 2; ++n) {                                                              \
            /* small terms first; every product is exact in fp32, the accumulation rounds like an fp32 sum */        \
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[AS][m][2], carry.b[BS][n][0], acc[m][n], 0, 0, 0); \
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[AS][m][1], carry.b[BS][n][1], acc[m][n], 0, 0, 0); \
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[AS][m][0], carry.b[BS][n][2], acc[m][n], 0, 0, 0); \
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[AS][m][1], carry.b[BS][n][0], acc[m][n], 0, 0, 0); \
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[AS][m][0], carry.b[BS][n][1], acc[m][n], 0, 0, 0); \
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[AS][m][0], carry.b[BS][n][0], acc[m][n], 0, 0, 0); \
        }
    // one MFMA, then a few VALU operations of the next block's split, ...
#define RS_INTERLEAVE()                                                                                              \
    _Pragma("unroll") for (int i = 0; i < NMT * 12; ++i) {                                                           \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                           \
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0); }
    RS_LOADA(0, ap)
    RS_LOADA(1, ap + 16)
    RS_SPLIT(0, 0)
    ap += 32;                                         // k-block 2
#pragma unroll 1
    for (int kb = 0; kb < KB16; kb += 2) {
        const bool more = kb + 2 < KB16;              // wave-uniform
        // ---- even block: a[0] x set 0; request set 1 <- block kb+1; raw[0] <- A block kb+2; split raw[1] -> a[1]
        split_load_set(q0, q1, carry.b[1]);
        if (more) { RS_LOADA(0, ap) }
        __builtin_amdgcn_sched_barrier(0);
        RS_SPLIT(1, 1)
        RS_MFMAS(0, 0)
        RS_INTERLEAVE()
        __builtin_amdgcn_sched_barrier(0);
        // ---- odd block: a[1] x set 1; request set 0 <- block kb+2 (or block 0 of the next GEMM); raw[1] <- A block kb+3
        q0 = more ? q0 + 192 : next.p;
        q1 = more ? q1 + 192 : next.p + next.ns;
        split_load_set(q0, q1, carry.b[0]);
        q0 += 192; q1 += 192;
        if (more) { RS_LOADA(1, ap + 16) }
        __builtin_amdgcn_sched_barrier(0);
        if (more) { RS_SPLIT(0, 0) }
        RS_MFMAS(1, 1)
        RS_INTERLEAVE()
        __builtin_amdgcn_sched_barrier(0);
        ap += 32;
    }
#undef RS_LOADA
#undef RS_SPLIT
#undef RS_MFMAS
#undef RS_INTERLEAVE
}

// ---------------------------------------------------------------------------------------------
// 16-row tiles: v_mfma_f32_16x16x32_bf16 (16 cycles per MFMA; six per fp32 product = 96 matrix cycles per 32 k-values
// and 16x16 tile against 256 for the eight v_mfma_f32_16x16x4_f32 they replace).  Same split, same fp32 LDS image, same
// accumulator layout as the fp32 16-row path (row = 4 (lane >> 4) + reg, col = lane & 15).
//   B (global): Ws16[((nt * KB32 + kb) * 3 + s) * 64 + lane] = 8 bf16 { W_s[o][k_j] }, o = 16 nt + (lane & 15), g = lane >> 4,
//               k_j = 32 kb + 4 g + j (j < 4) and 32 kb + 16 + 4 g + (j - 4) (j >= 4)
//   A (LDS):    the lane reads two float4 of row (lane & 15) at k = 32 kb + 4 g and 32 kb + 16 + 4 g - the conflict-free
//               ds_read_b128 pattern of the fp32 16-row path, twice (the MFMA's k order inside a block is free as long as A and B agree)
// The weight stream is what binds a 16-row tile (one workgroup streams every weight of the block for 16 rows: 6 B per
// weight here against 4 B on the fp32 instruction), so fragments are requested THREE k-blocks ahead of their use (a ring
// of four register sets, 36 KB in flight per wave); the kernel needs the 512-register budget of one wave per SIMD.
// ---------------------------------------------------------------------------------------------
struct S16FragPtr { const sbf16x8* p; unsigned ns; };   // a wave's first 16-column tile; ns = stride between n-tiles (16-byte units)
__device__ __forceinline__ S16FragPtr sfrag16_ptr(const void* Ws16, int kb32_total, int kb0, int cg) {
    const int lane = threadIdx.x & 63;
    S16FragPtr f;
    f.p = reinterpret_cast<const sbf16x8*>(Ws16) + ((size_t)(4 * cg) * kb32_total + kb0) * 192 + lane;
    f.ns = (unsigned)kb32_total * 192u;
    return f;
}
struct S16Carry { sbf16x8 b[4][4][3]; };                // ring of four k-blocks x [4 n-tiles][3 pieces]
__device__ __forceinline__ void split16_load_set(const sbf16x8* q, unsigned ns, sbf16x8 (&dst)[4][3]) {
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int s = 0; s < 3; ++s) dst[n][s] = q[n * ns + s * 64];
}
// a GEMM enters with its k-blocks 0, 1, 2 in sets 0, 1, 2 and leaves with those of `next` there
__device__ __forceinline__ void split16_prefetch(const S16FragPtr& f, S16Carry& c) {
    split16_load_set(f.p, f.ns, c.b[0]); split16_load_set(f.p + 192, f.ns, c.b[1]); split16_load_set(f.p + 384, f.ns, c.b[2]);
}

typedef float sf32x4 __attribute__((ext_vector_type(4)));
template <int KB32>
__device__ __forceinline__ void tile_gemm_rsplit16(const float* ldsA, int lda, const S16FragPtr cur, const S16FragPtr next,
                                                   sf32x4 (&acc)[4], S16Carry& carry) {
    static_assert(KB32 % 4 == 0, "K must be a multiple of 128");
    const int lane = threadIdx.x & 63;
    const float* ap = ldsA + (lane & 15) * lda + (lane >> 4) * 4;
    float4 raw[2][2];
    sbf16x8 a[2][3];
#define R16_LOADA(SET, PTR) { raw[SET][0] = *reinterpret_cast<const float4*>(PTR); raw[SET][1] = *reinterpret_cast<const float4*>((PTR) + 16); }
#define R16_SPLIT(DST, SET) split8(raw[SET][0], raw[SET][1], a[DST][0], a[DST][1], a[DST][2]);
    // small terms first, n-tiles interleaved so that consecutive MFMAs never share an accumulator
#define R16_MFMAS(AS, BS)                                                                                                   \
    _Pragma("unroll") for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[AS][2], carry.b[BS][n][0], acc[n], 0, 0, 0); \
    _Pragma("unroll") for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[AS][1], carry.b[BS][n][1], acc[n], 0, 0, 0); \
    _Pragma("unroll") for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[AS][0], carry.b[BS][n][2], acc[n], 0, 0, 0); \
    _Pragma("unroll") for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[AS][1], carry.b[BS][n][0], acc[n], 0, 0, 0); \
    _Pragma("unroll") for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[AS][0], carry.b[BS][n][1], acc[n], 0, 0, 0); \
    _Pragma("unroll") for (int n = 0; n < 4; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[AS][0], carry.b[BS][n][0], acc[n], 0, 0, 0);
    // the 16-cycle MFMA holds the vector issue port for 8 cycles: two VALU operations of the next block's split fit each gap
#define R16_INTERLEAVE()                                                                                                    \
    _Pragma("unroll") for (int i = 0; i < 24; ++i) {                                                                        \
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                                  \
        __builtin_amdgcn_sched_group_barrier(0x002, 2, 0); }
    // block i: MFMAs on set i % 4; set (i + 3) % 4 <- weight block i + 3 (or block i + 3 - KB32 of `next`);
    // raw[i % 2] <- A block i + 2; a[(i + 1) % 2] <- split of raw[(i + 1) % 2]
#define R16_BLOCK(I)                                                                                                        \
    {                                                                                                                       \
        const bool tail = kb + (I) + 3 >= KB32;                      /* wave-uniform */                                     \
        const sbf16x8* q = tail ? next.p + (unsigned)(kb + (I) + 3 - KB32) * 192u : cur.p + (unsigned)(kb + (I) + 3) * 192u; \
        split16_load_set(q, tail ? next.ns : cur.ns, carry.b[((I) + 3) & 3]);                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                                  \
        if (kb + (I) + 1 < KB32) { R16_SPLIT(((I) + 1) & 1, ((I) + 1) & 1) }                                                \
        R16_MFMAS((I) & 1, (I) & 3)                                                                                         \
        R16_INTERLEAVE()                                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                                  \
        if (kb + (I) + 2 < KB32) { R16_LOADA((I) & 1, ap + (kb + (I) + 2) * 32) }                                           \
    }
    R16_LOADA(0, ap)
    R16_LOADA(1, ap + 32)
    R16_SPLIT(0, 0)
#pragma unroll 1
    for (int kb = 0; kb < KB32; kb += 4) {
        R16_BLOCK(0) R16_BLOCK(1) R16_BLOCK(2) R16_BLOCK(3)
    }
#undef R16_LOADA
#undef R16_SPLIT
#undef R16_MFMAS
#undef R16_INTERLEAVE
#undef R16_BLOCK
}

// ---------------------------------------------------------------------------------------------
// Plane variant for tiles whose PRODUCER can split: the A operand lies in LDS as three bf16 planes
// [rows][KH + 8] (KH = the k-range held at a time), written once per element by whoever builds the tile
// (split_store4), so the GEMM loop carries no VALU work at all.  Used by the edge kernels with KH = K/2: two
// half-K passes keep the tile at 52 KB (inside the 66 KB fp32 image it aliases), two workgroups per CU.
// ---------------------------------------------------------------------------------------------
#define SPLIT_PLANE_LDA(KH) ((KH) + 8)            // bf16 elements per plane row: row stride (KH*2 + 16) bytes, conflict-free ds_read_b128

// four consecutive k-values of one row -> the three planes (8 bytes each)
__device__ __forceinline__ void split_store4(unsigned short* planes, int plane_elems, int off, const float4& v) {
    uint32_t a0, a1, a2, b0, b1, b2;
    split3_pair(v.x, v.y, a0, a1, a2);
    split3_pair(v.z, v.w, b0, b1, b2);
    *reinterpret_cast<uint2*>(planes + off) = make_uint2(a0, b0);
    *reinterpret_cast<uint2*>(planes + plane_elems + off) = make_uint2(a1, b1);
    *reinterpret_cast<uint2*>(planes + 2 * plane_elems + off) = make_uint2(a2, b2);
}

#ifndef CMDGEN_PLANE_PIN
#define CMDGEN_PLANE_PIN 1      // tile_gemm_planes: one load pinned after every one or two MFMAs (0: loads in a burst per k-block)
#endif
// NPC = 1: only the leading piece of both operands (plain bf16 operands, fp32 accumulation - the training step's opt-in mixed precision)
template <int MT, int KB16, int NPC = 3>
__device__ __forceinline__ void tile_gemm_planes(const unsigned short* planes, int plane_elems, int lda,
                                                 const SFragPtr cur, const SFragPtr next,
                                                 sf32x16 (&acc)[MT / 32][2], SCarry& carry) {
    static_assert(KB16 % 2 == 0, "the k-range of a pass must be a multiple of 32");
    constexpr int NMT = MT / 32;
    const int lane = threadIdx.x & 63;
    const unsigned short* ap = planes + (lane & 31) * lda + (lane >> 5) * 8;
    const sbf16x8* q0 = cur.p + 192;
    const sbf16x8* q1 = q0 + cur.ns;
    sbf16x8 a[2][NMT][3];
#define PL_LOADA(SET, PTR)                                                                                           \
    _Pragma("unroll") for (int m = 0; m < NMT; ++m)                                                                  \
        _Pragma("unroll") for (int s = 0; s < NPC; ++s)                                                              \
            a[SET][m][s] = *reinterpret_cast<const sbf16x8*>((PTR) + s * plane_elems + m * 32 * lda);
#define PL_MFMAS(AS, BS)                                                                                             \
    _Pragma("unroll") for (int m = 0; m < NMT; ++m)                                                                  \
        _Pragma("unroll") for (int n = 0; n < 2; ++n) {                                                              \
            if constexpr (NPC == 3) {                                                                                \
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[AS][m][2], carry.b[BS][n][0], acc[m][n], 0, 0, 0); \
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[AS][m][1], carry.b[BS][n][1], acc[m][n], 0, 0, 0); \
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[AS][m][0], carry.b[BS][n][2], acc[m][n], 0, 0, 0); \
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[AS][m][1], carry.b[BS][n][0], acc[m][n], 0, 0, 0); \
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[AS][m][0], carry.b[BS][n][1], acc[m][n], 0, 0, 0); \
            }                                                                                                        \
            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[AS][m][0], carry.b[BS][n][0], acc[m][n], 0, 0, 0); \
        }
    PL_LOADA(0, ap)
    if constexpr (NPC == 3 && CMDGEN_PLANE_PIN != 0) {
        // One load pinned in the shadow of every one or two MFMAs (sched_barrier after each group) instead of a burst of 9-12 loads and
        // their address arithmetic between two k-blocks, which left the matrix pipe idle ~100 cycles per block (profiles/r03_m_node64.txt).
        // Per accumulator the order of the six products is unchanged (small terms first).
#define PL_LA(SET, PTR, M, S) a[SET][M][S] = *reinterpret_cast<const sbf16x8*>((PTR) + (S) * plane_elems + (M) * 32 * lda);
#define PL_LB(SET, N, S) carry.b[SET][N][S] = ((N) == 0 ? q0 : q1)[(S) * 64];
#define PL_MF(M, N, AS, AI, BS, BI) acc[M][N] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[AS][M][AI], carry.b[BS][N][BI], acc[M][N], 0, 0, 0);
#define PL_GRP(LOADS, M, AS, BS, AI, BI) LOADS PL_MF(M, 0, AS, AI, BS, BI) PL_MF(M, 1, AS, AI, BS, BI) __builtin_amdgcn_sched_barrier(0);
#define PL_HALF(AS, BS, AN, BN, PTR)                                                                                              \
        if constexpr (NMT == 2) {                                                                                                 \
            PL_GRP(PL_LA(AN, PTR, 0, 2), 0, AS, BS, 2, 0) PL_GRP(PL_LA(AN, PTR, 1, 2), 1, AS, BS, 2, 0)                             \
            PL_GRP(PL_LA(AN, PTR, 0, 1), 0, AS, BS, 1, 1) PL_GRP(PL_LA(AN, PTR, 1, 1), 1, AS, BS, 1, 1)                             \
            PL_GRP(PL_LA(AN, PTR, 0, 0), 0, AS, BS, 0, 2) PL_GRP(PL_LA(AN, PTR, 1, 0), 1, AS, BS, 0, 2)                             \
            PL_GRP(PL_LB(BN, 0, 0), 0, AS, BS, 1, 0) PL_GRP(PL_LB(BN, 0, 1), 1, AS, BS, 1, 0)                                       \
            PL_GRP(PL_LB(BN, 0, 2), 0, AS, BS, 0, 1) PL_GRP(PL_LB(BN, 1, 0), 1, AS, BS, 0, 1)                                       \
            PL_GRP(PL_LB(BN, 1, 1), 0, AS, BS, 0, 0) PL_GRP(PL_LB(BN, 1, 2), 1, AS, BS, 0, 0)                                       \
        } else {                                                                                                                  \
            PL_GRP(PL_LA(AN, PTR, 0, 2) PL_LB(BN, 0, 0), 0, AS, BS, 2, 0) PL_GRP(PL_LA(AN, PTR, 0, 1) PL_LB(BN, 0, 1), 0, AS, BS, 1, 1) \
            PL_GRP(PL_LA(AN, PTR, 0, 0) PL_LB(BN, 0, 2), 0, AS, BS, 0, 2) PL_GRP(PL_LB(BN, 1, 0), 0, AS, BS, 1, 0)                  \
            PL_GRP(PL_LB(BN, 1, 1), 0, AS, BS, 0, 1) PL_GRP(PL_LB(BN, 1, 2), 0, AS, BS, 0, 0)                                       \
        }
#pragma unroll 1
        for (int kb = 0; kb < KB16; kb += 2) {
            const bool more = kb + 2 < KB16;
            const unsigned short* an = more ? ap + 32 : ap;          // (last block: re-reads its own fragments, unused)
            __builtin_amdgcn_sched_barrier(0);
            PL_HALF(0, 0, 1, 1, ap + 16)
            q0 = more ? q0 + 192 : next.p;
            q1 = more ? q1 + 192 : next.p + next.ns;
            __builtin_amdgcn_sched_barrier(0);
            PL_HALF(1, 1, 0, 0, an)
            q0 += 192; q1 += 192;
            ap += 32;
        }
#undef PL_LA
#undef PL_LB
#undef PL_MF
#undef PL_GRP
#undef PL_HALF
    } else {
#pragma unroll 1
    for (int kb = 0; kb < KB16; kb += 2) {
        const bool more = kb + 2 < KB16;
        split_load_set<NPC>(q0, q1, carry.b[1]);
        PL_LOADA(1, ap + 16)
        __builtin_amdgcn_sched_barrier(0);
        PL_MFMAS(0, 0)
        __builtin_amdgcn_sched_barrier(0);
        q0 = more ? q0 + 192 : next.p;
        q1 = more ? q1 + 192 : next.p + next.ns;
        split_load_set<NPC>(q0, q1, carry.b[0]);
        q0 += 192; q1 += 192;
        if (more) { PL_LOADA(0, ap + 32) }
        __builtin_amdgcn_sched_barrier(0);
        PL_MFMAS(1, 1)
        __builtin_amdgcn_sched_barrier(0);
        ap += 32;
    }
    }
#undef PL_LOADA
#undef PL_MFMAS
}

// ---------------------------------------------------------------------------------------------
// Full-K plane image for 32-row tiles (round 3): three planes of [32][256] bf16 with NO row pad (48 KB: three workgroups per CU) - the
// 16-byte k-octet o of row r lies at octet o ^ (r & 31) of the row, so the sixteen rows of a ds_read_b128 phase fall on sixteen different
// bank groups.  One build -> one GEMM over all 256 k-values per tile instead of two half-K passes: the tile's gathers are one round trip
// instead of two (at 64 pockets a launch is one tile's latency chain, and the second gather was 2 us of it; profiles/r03_t_fullk.txt).
// Carry contract as tile_gemm_planes (set 0 holds k-block 0 of `cur` on entry, of `next` on exit); loads pinned one by one under the MFMAs.
// ---------------------------------------------------------------------------------------------
#define SPLIT_SWZ_PE (32 * 256)                               // bf16 elements per plane
__device__ __forceinline__ int split_swz_off(int row, int octet) { return row * 256 + ((octet ^ (row & 31)) << 3); }

// four consecutive k-values (4 c4 .. 4 c4 + 3) of row e -> the three swizzled planes
__device__ __forceinline__ void split_store4_swz(unsigned short* planes, int e, int c4, const float4& v) {
    split_store4(planes, SPLIT_SWZ_PE, split_swz_off(e, c4 >> 1) + ((c4 & 1) << 2), v);
}

__device__ __forceinline__ void tile_gemm_planes_swz32(const unsigned short* planes, const SFragPtr cur, const SFragPtr next,
                                                       sf32x16 (&acc)[1][2], SCarry& carry) {
    constexpr int KB16 = 16, PE = SPLIT_SWZ_PE;
    const int lane = threadIdx.x & 63, row = lane & 31, hi = lane >> 5;
    const unsigned short* rp = planes + row * 256;
    const sbf16x8* q0 = cur.p + 192;
    const sbf16x8* q1 = q0 + cur.ns;
    sbf16x8 a[2][3];
#define SW_PTR(KB) (rp + (((2 * (KB) + hi) ^ row) << 3))
#define SW_LA(SET, PTR, S) a[SET][S] = *reinterpret_cast<const sbf16x8*>((PTR) + (S) * PE);
#define SW_LB(SET, N, S) carry.b[SET][N][S] = ((N) == 0 ? q0 : q1)[(S) * 64];
#define SW_MF(N, AS, AI, BS, BI) acc[0][N] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[AS][AI], carry.b[BS][N][BI], acc[0][N], 0, 0, 0);
#define SW_GRP(LOADS, AS, BS, AI, BI) LOADS SW_MF(0, AS, AI, BS, BI) SW_MF(1, AS, AI, BS, BI) __builtin_amdgcn_sched_barrier(0);
#define SW_HALF(AS, BS, AN, BN, PTR)                                                                                              \
        SW_GRP(SW_LA(AN, PTR, 2) SW_LB(BN, 0, 0), AS, BS, 2, 0) SW_GRP(SW_LA(AN, PTR, 1) SW_LB(BN, 0, 1), AS, BS, 1, 1)             \
        SW_GRP(SW_LA(AN, PTR, 0) SW_LB(BN, 0, 2), AS, BS, 0, 2) SW_GRP(SW_LB(BN, 1, 0), AS, BS, 1, 0)                               \
        SW_GRP(SW_LB(BN, 1, 1), AS, BS, 0, 1) SW_GRP(SW_LB(BN, 1, 2), AS, BS, 0, 0)
    { const unsigned short* p0 = SW_PTR(0); SW_LA(0, p0, 0) SW_LA(0, p0, 1) SW_LA(0, p0, 2) }
#pragma unroll 1
    for (int kb = 0; kb < KB16; kb += 2) {
        const bool more = kb + 2 < KB16;
        const unsigned short* p1 = SW_PTR(kb + 1);
        const unsigned short* p2 = SW_PTR(more ? kb + 2 : kb);       // (last block: re-reads its own fragments, unused)
        __builtin_amdgcn_sched_barrier(0);
        SW_HALF(0, 0, 1, 1, p1)
        q0 = more ? q0 + 192 : next.p;
        q1 = more ? q1 + 192 : next.p + next.ns;
        __builtin_amdgcn_sched_barrier(0);
        SW_HALF(1, 1, 0, 0, p2)
        q0 += 192; q1 += 192;
    }
#undef SW_PTR
#undef SW_LA
#undef SW_LB
#undef SW_MF
#undef SW_GRP
#undef SW_HALF
}

// ---------------------------------------------------------------------------------------------
// Half engine (two fp16 pieces per operand, three MFMAs per product; see the top of this file) for the full-K 32-row plane image:
// two swizzled planes of [32][256] fp16 (32 KB), weight fragments from WPack::wh ([nt][K/16][2 pieces][64 lanes] x 16 bytes).
// Carry contract as tile_gemm_planes_swz32.  The accumulators carry WPack::wh_scale.
// ---------------------------------------------------------------------------------------------
struct HFragPtr { const sf16x8* p; unsigned ns; };      // a wave's first n-tile; ns = stride between n-tiles (16-byte units)
__device__ __forceinline__ HFragPtr hfrag_ptr(const void* Wh, int kb16_total, int kb0, int cg) {
    const int lane = threadIdx.x & 63;
    HFragPtr f;
    f.p = reinterpret_cast<const sf16x8*>(Wh) + ((size_t)(2 * cg) * kb16_total + kb0) * 128 + lane;
    f.ns = (unsigned)kb16_total * 128u;
    return f;
}
struct HCarry { sf16x8 b[2][2][2]; };                   // two register sets of [2 n-tiles][2 pieces]
__device__ __forceinline__ void half_prefetch(const HFragPtr& f, HCarry& c) {
#pragma unroll
    for (int s = 0; s < 2; ++s) { c.b[0][0][s] = f.p[s * 64]; c.b[0][1][s] = (f.p + f.ns)[s * 64]; }
}
// four consecutive k-values (4 c4 .. 4 c4 + 3) of row e -> the two swizzled fp16 planes
__device__ __forceinline__ void split_store4_swz_half(unsigned short* planes, int e, int c4, const float4& v) {
    split_store4_half(planes, SPLIT_SWZ_PE, split_swz_off(e, c4 >> 1) + ((c4 & 1) << 2), v);
}
__device__ __forceinline__ void tile_gemm_planes_swz32_half(const unsigned short* planes, const HFragPtr cur, const HFragPtr next,
                                                            sf32x16 (&acc)[1][2], HCarry& carry) {
    constexpr int KB16 = 16, PE = SPLIT_SWZ_PE;
    const int lane = threadIdx.x & 63, row = lane & 31, hi = lane >> 5;
    const unsigned short* rp = planes + row * 256;
    const sf16x8* q0 = cur.p + 128;
    const sf16x8* q1 = q0 + cur.ns;
    sf16x8 a[2][2];
#define HW_PTR(KB) (rp + (((2 * (KB) + hi) ^ row) << 3))
#define HW_LA(SET, PTR, S) a[SET][S] = *reinterpret_cast<const sf16x8*>((PTR) + (S) * PE);
#define HW_LB(SET, N, S) carry.b[SET][N][S] = ((N) == 0 ? q0 : q1)[(S) * 64];
#define HW_MF(N, AS, AI, BS, BI) acc[0][N] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[AS][AI], carry.b[BS][N][BI], acc[0][N], 0, 0, 0);
#define HW_GRP(LOADS, AS, BS, AI, BI) LOADS HW_MF(0, AS, AI, BS, BI) HW_MF(1, AS, AI, BS, BI) __builtin_amdgcn_sched_barrier(0);
    // a1 b0, a0 b1, a0 b0 (small terms first); two loads in the shadow of every two MFMAs
#define HW_HALF(AS, BS, AN, BN, PTR)                                                                                              \
        HW_GRP(HW_LA(AN, PTR, 1) HW_LB(BN, 0, 0), AS, BS, 1, 0) HW_GRP(HW_LA(AN, PTR, 0) HW_LB(BN, 0, 1), AS, BS, 0, 1)             \
        HW_GRP(HW_LB(BN, 1, 0) HW_LB(BN, 1, 1), AS, BS, 0, 0)
    { const unsigned short* p0 = HW_PTR(0); HW_LA(0, p0, 0) HW_LA(0, p0, 1) }
#pragma unroll 1
    for (int kb = 0; kb < KB16; kb += 2) {
        const bool more = kb + 2 < KB16;
        const unsigned short* p1 = HW_PTR(kb + 1);
        const unsigned short* p2 = HW_PTR(more ? kb + 2 : kb);       // (last block: re-reads its own fragments, unused)
        __builtin_amdgcn_sched_barrier(0);
        HW_HALF(0, 0, 1, 1, p1)
        q0 = more ? q0 + 128 : next.p;
        q1 = more ? q1 + 128 : next.p + next.ns;
        __builtin_amdgcn_sched_barrier(0);
        HW_HALF(1, 1, 0, 0, p2)
        q0 += 128; q1 += 128;
    }
#undef HW_PTR
#undef HW_LA
#undef HW_LB
#undef HW_MF
#undef HW_GRP
#undef HW_HALF
}
