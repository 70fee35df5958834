// fq2p.h -- Fq2 arithmetic with one element spread over a PAIR of adjacent lanes (device only).
//
// A G2 mixed addition on one lane keeps ~11 Fq2 values (24 registers each) live and spills ~2000 dwords on
// gfx950 at any occupancy.  Here lane 2k holds the c0 halves and lane 2k+1 the c1 halves of every Fq2 value, so
// a lane's register footprint equals the G1 kernel's.  Halves travel between the two lanes with DPP
// quad_perm [1,0,3,2] moves (one VALU instruction per limb, no LDS).
//   mul : 4 Fq multiplications per pair (2 per lane, schoolbook):  c0 = a0 b0 + beta a1 b1,  c1 = a1 b0 + a0 b1
//   sqr : 2 per pair (1 per lane):  lane0 (a0 - a1)(a0 - beta a1),  lane1 a0 a1   (quadratic_extension.rs:257-305)
// so a mixed addition (7 M + 4 S) costs 18 multiplication-times per lane pair = 36 lane-multiplications,
// against 33 for the single-lane Karatsuba form -- and none of them spill.
// Values are the same field elements as Fq2's (fields/models/quadratic_extension.rs), beta = -5.
#pragma once
#include "curve.h"

namespace czk {

__device__ __forceinline__ u32 pair_swap_u32(u32 x) {
    return (u32)__builtin_amdgcn_mov_dpp((int)x, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true);
}
__device__ __forceinline__ Fq pair_swap(const Fq& a) {
    Fq r;
#pragma unroll
    for (int i = 0; i < 12; i++) r.l[i] = pair_swap_u32(a.l[i]);
    return r;
}
__device__ __forceinline__ bool pair_parity() { return (threadIdx.x & 1u) != 0; }
__device__ __forceinline__ Fq fq_select(bool c, const Fq& a, const Fq& b) {   // c ? a : b
    Fq r;
#pragma unroll
    for (int i = 0; i < 12; i++) r.l[i] = c ? a.l[i] : b.l[i];
    return r;
}

struct alignas(16) Fq2P {
    Fq h;   // this lane's half: c0 on even lanes, c1 on odd lanes
    static __device__ __forceinline__ Fq2P zero() { return Fq2P{Fq::zero()}; }
    static __device__ __forceinline__ Fq2P one() { return Fq2P{pair_parity() ? Fq::zero() : Fq::one()}; }
    __device__ __forceinline__ bool is_zero() const {
        u32 z = h.is_zero() ? 1u : 0u;
        return (z & pair_swap_u32(z)) != 0;
    }
    __device__ __forceinline__ bool operator==(const Fq2P& b) const {
        u32 e = (h == b.h) ? 1u : 0u;
        return (e & pair_swap_u32(e)) != 0;
    }
    __device__ __forceinline__ bool operator!=(const Fq2P& b) const { return !(*this == b); }
};

__device__ __forceinline__ Fq2P f_add(const Fq2P& a, const Fq2P& b) { return Fq2P{fp_add(a.h, b.h)}; }
__device__ __forceinline__ Fq2P f_sub(const Fq2P& a, const Fq2P& b) { return Fq2P{fp_sub(a.h, b.h)}; }
__device__ __forceinline__ Fq2P f_dbl(const Fq2P& a) { return Fq2P{fp_dbl(a.h)}; }
__device__ __forceinline__ Fq2P f_neg(const Fq2P& a) { return Fq2P{fp_neg(a.h)}; }

__device__ __forceinline__ Fq2P f_mul(const Fq2P& a, const Fq2P& b) {
    const bool par = pair_parity();
    Fq pa = pair_swap(a.h), pb = pair_swap(b.h);
    // even lane: t1 = a0 b0, t2 = a1 b1      odd lane: t1 = a1 b0, t2 = a0 b1
    Fq t1 = fp_mul(a.h, fq_select(par, pb, b.h));
    Fq t2 = fp_mul(pa, fq_select(par, b.h, pb));
    return Fq2P{fp_add(t1, fq_select(par, t2, fq_mul_by_nonresidue(t2)))};
}
__device__ __forceinline__ Fq2P f_sqr(const Fq2P& a) {
    const bool par = pair_parity();
    Fq pa = pair_swap(a.h);   // even lane: a1, odd lane: a0
    // even lane: (a0 - a1) * (a0 - beta a1)      odd lane: a0 * a1
    Fq x = fq_select(par, pa, fp_sub(a.h, pa));
    Fq y = fq_select(par, a.h, fp_sub(a.h, fq_mul_by_nonresidue(pa)));
    Fq t = fp_mul(x, y);
    Fq pt = pair_swap(t);     // even lane receives v2 = a0 a1
    // c0 = v0 v3 + v2 + beta v2      c1 = 2 v2
    Fq even = fp_add(fp_add(t, pt), fq_mul_by_nonresidue(pt));
    return Fq2P{fq_select(par, fp_dbl(t), even)};
}

// memory layout of an Fq2 is (c0, c1): this lane touches words [6*parity, 6*parity + 6)
template <>
struct FieldIO<Fq2P> {
    static constexpr int W64 = 12;
    static __device__ __forceinline__ Fq2P load(const u64* p) { return Fq2P{fp_load<FqParams>(p + (pair_parity() ? 6 : 0))}; }
    static __device__ __forceinline__ void store(u64* p, const Fq2P& a) { fp_store<FqParams>(p + (pair_parity() ? 6 : 0), a.h); }
};

}  // namespace czk
