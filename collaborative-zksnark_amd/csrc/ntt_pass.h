// ntt_pass.h -- interface between the NTT driver (ntt.hip) and the second-generation pass kernels (ntt_pass.hip)
#pragma once
#include "czk_internal.h"
#include "fru.h"

namespace czk {

// Format of the scratch lanes between passes:
//   0  canonical elements (8 x u32, < r);
//   1  LAZY elements of nine 29-bit limbs = 36 bytes, value < 2 r: saves the pack / conditional subtraction / unpack round trip but
//      reads and writes 36-byte elements with 4-byte accesses -- measured SLOWER on MI355X (2^21 x 4 lanes: 0.914 ms against 0.881 ms);
//   2  (default) values < 2 r PACKED into the same 8 x u32 as format 0 (2 r < 2^254): 16-byte accesses like format 0, but the
//      conditional subtraction that makes the value canonical is only done by the last pass.
// In formats 1 and 2 a pass after the first starts from the bound 2 instead of 1.003 and uses the next larger constants (see the
// bound analysis in ntt_pass.hip).  All three are kept compilable; tests run against whichever is selected.
constexpr int NTT2_SCRATCH_FORMAT = 2;
constexpr bool NTT2_LAZY_SCRATCH = NTT2_SCRATCH_FORMAT == 1;
constexpr bool NTT2_SCRATCH_2R = NTT2_SCRATCH_FORMAT != 0;
constexpr size_t NTT2_SCRATCH_ELEM_BYTES = NTT2_LAZY_SCRATCH ? 36 : 32;

struct Pass2Args {
    const u64* in;        // first pass: the caller's lanes (canonical 8 x u32 per element); later passes: the scratch lanes
    u64* out;             //   (see NTT2_LAZY_SCRATCH); the last pass writes canonical elements to the caller's lanes
    const u32* tw;        // per-stage compacted twiddles, unsaturated form: entry (2^s - 1 + j) = 9 limbs of w_s^j 2^261 mod r
    const u32* prescale;  // g^i 2^261 (first pass of a coset fft) or null
    const u32* posttab;   // size_inv g^-i 2^261 (last pass of a coset ifft) or null
    FrU postconst;        // size_inv 2^261 (last pass of an ifft)
    int post_mode;        // 0 none, 1 constant, 2 table
    unsigned n;           // log2 D
    unsigned s_lo;        // lowest stage bit of this pass
    size_t in_len;        // elements >= in_len read as zero (only honoured when `first`)
    int first;
    size_t lane_stride;   // elements between lanes (= D)
    size_t in_lane_stride;   // ... of `in` in the first pass (out-of-place transforms read the caller's source lanes)
};

int launch_ntt2_pass(czk_ctx* ctx, const Pass2Args& a, unsigned K, bool last, size_t lanes);
int launch_ntt2_final_first(czk_ctx* ctx, const Pass2Args& ai, const Pass2Args& af, unsigned C, size_t lanes);
void launch_table_to_u(hipStream_t st, const u64* sat, size_t count, u32* dst);
FrU host_fr_to_u(const Fr& sat);

}  // namespace czk
