// poly.hip -- the share-local callers on either side of the NTT / MSM hot path ("next" rows of the scope table):
//   * R1CS constraint evaluation  a_i = <A_i, z>   (mpc-snarks/src/groth/r1cs_to_qap.rs:12-42, 67-83, 95-100)
//   * division of a coefficient vector by (X - z)  (KZG10::compute_witness_polynomial, poly-commit/src/kzg10/mod.rs:200-224;
//     on shares: mpc-algebra/src/share/add.rs:148-156 -- the divisor is public, so the division is lane-wise)
// Both are one- or two-sweep HBM-bound Fr kernels; share lanes ride on gridDim.y.
#include "czk_internal.h"

#include <string.h>

namespace czk {

__device__ __forceinline__ Fr pfr_load(const u64* base, size_t idx) { return fp_load<FrParams>(base + 4 * idx); }
__device__ __forceinline__ void pfr_store(u64* base, size_t idx, const Fr& v) { fp_store<FrParams>(base + 4 * idx, v); }

// ------------------------------------------------------------------------------------------------
// R1CS matrices in CSR.  col[t] bit 31 marks a coefficient equal to one: evaluate_constraint adds the variable
// without multiplying (r1cs_to_qap.rs:28-32) and the kernel then skips the 32-byte coefficient read as well.
// ------------------------------------------------------------------------------------------------
__global__ void k_r1cs_prepare(const u64* row_ptr, u32* col, const u64* coeff, size_t m, size_t nnz, size_t n_vars, u32* row_ptr32, u32* bad) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t <= m) {
        u64 r = row_ptr[t];
        if (r > nnz || (t && row_ptr[t - 1] > r) || (t == 0 && r != 0) || (t == m && r != nnz)) atomicOr(bad, 1u);
        row_ptr32[t] = (u32)r;
    }
    if (t < nnz) {
        u32 c = col[t];
        if (c >= n_vars || c >= 0x80000000u) {
            atomicOr(bad, 2u);
            return;
        }
        Fr k = pfr_load(coeff, t);
        const Fr one = Fr::one();
        bool is_one = true;
#pragma unroll
        for (int i = 0; i < 8; i++) is_one = is_one && (k.l[i] == one.l[i]);
        if (is_one) col[t] = c | 0x80000000u;
    }
}

// out[lane][i] = sum_t coeff[t] * z[lane][col[t]] over row i; one thread per row
__global__ __launch_bounds__(256) void k_r1cs_matvec(const u32* row_ptr, const u32* col, const u64* coeff, size_t m, const u64* z, size_t z_stride,
                                                      u64* out, size_t out_stride) {
    const unsigned lane = blockIdx.y;
    const u64* zl = z + 4 * (size_t)lane * z_stride;
    u64* ol = out + 4 * (size_t)lane * out_stride;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < m; i += (size_t)gridDim.x * blockDim.x) {
        u32 t0 = row_ptr[i], t1 = row_ptr[i + 1];
        Fr acc = Fr::zero();
        for (u32 t = t0; t < t1; t++) {
            u32 c = col[t];
            Fr v = pfr_load(zl, c & 0x7fffffffu);
            if (!(c & 0x80000000u)) v = fp_mul(v, pfr_load(coeff, t));
            acc = fp_add(acc, v);
        }
        pfr_store(ol, i, acc);
    }
}

// ------------------------------------------------------------------------------------------------
// Suffix Horner sums R_i = sum_{j >= i} p_j x^(j - i) (so R_i = p_i + x R_{i+1}): the quotient of p / (X - x) is
// q_k = R_{k+1} and the remainder R_0 = p(x).  Segments of SEG coefficients: (1) per-segment Horner value H_t;
// (2) the H_t are themselves a coefficient vector whose suffix Horner sums at x^SEG are the carries into the
// segments -- the same problem, SEG times smaller (recursion, <= 3 levels up to 2^21); (3) per-segment fill.
// ------------------------------------------------------------------------------------------------
constexpr unsigned SEG = 32;

__global__ __launch_bounds__(256) void k_seg_horner(const u64* in, size_t in_stride, size_t n, Fr x, u64* H, size_t n_seg) {
    const unsigned lane = blockIdx.y;
    const u64* p = in + 4 * (size_t)lane * in_stride;
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_seg) return;
    size_t a = t * SEG, b = a + SEG < n ? a + SEG : n;
    Fr r = Fr::zero();
    for (size_t i = b; i-- > a;) r = fp_add(fp_mul(r, x), pfr_load(p, i));
    pfr_store(H + 4 * (size_t)lane * n_seg, t, r);
}
// out[i - shift] = R_i for i >= shift (shift = 1 writes the quotient); R_0 goes to rem when shift == 1.
// carry[t] = suffix sum entering segment t from above = R at index (t + 1) * SEG, i.e. element t + 1 of the next level.
__global__ __launch_bounds__(256) void k_seg_fill(const u64* in, size_t in_stride, size_t n, Fr x, const u64* carry, size_t n_seg, u64* out,
                                                  size_t out_stride, unsigned shift, u64* rem) {
    const unsigned lane = blockIdx.y;
    const u64* p = in + 4 * (size_t)lane * in_stride;
    u64* o = out + 4 * (size_t)lane * out_stride;
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_seg) return;
    size_t a = t * SEG, b = a + SEG < n ? a + SEG : n;
    Fr r = (carry && t + 1 < n_seg) ? pfr_load(carry + 4 * (size_t)lane * n_seg, t + 1) : Fr::zero();
    for (size_t i = b; i-- > a;) {
        r = fp_add(fp_mul(r, x), pfr_load(p, i));
        if (i >= shift) pfr_store(o, i - shift, r);
        else if (rem) pfr_store(rem, lane, r);
    }
}

// ------------------------------------------------------------------------------------------------
// Division by the vanishing polynomial X^n - 1 in coefficient form: a = q (X^n - 1) + r with
// q_i = sum_{k >= 1} a_{i + k n} and r_i = sum_{k >= 0} a_{i + k n} (i < n): the suffix sums of the
// n-coefficient chunks of a.  One thread per residue i < n walks its chunks from the top: every
// coefficient is read once, every output written once, neighbouring threads touch neighbouring elements.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_div_vanishing(const u64* in, size_t m, size_t n, u64* q, u64* rem) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned lane = blockIdx.y;
    const u64* a = in + 4 * (size_t)lane * m;
    u64* qo = q + 4 * (size_t)lane * (m - n);
    const size_t chunks = (m + n - 1) / n;
    Fr acc = Fr::zero();
    for (size_t k = chunks; k-- > 1;) {
        if (i + k * n < m) acc = fp_add(acc, pfr_load(a, i + k * n));
        if (i + (k - 1) * n < m - n) pfr_store(qo, i + (k - 1) * n, acc);
    }
    if (rem) pfr_store(rem + 4 * (size_t)lane * n, i, i < m ? fp_add(acc, pfr_load(a, i)) : acc);
}

static Fr host_pow(Fr x, unsigned e) {
    Fr r = Fr::one();
    for (; e; e >>= 1) {
        if (e & 1) r = fp_mul(r, x);
        x = fp_mul(x, x);
    }
    return r;
}

// suffix Horner sums of `in` (lanes x n, stride in_stride) at x, written to out[i - shift]; `ws` is a bump pointer into
// ctx->poly_scratch (suffix_horner_scratch bytes)
static size_t suffix_horner_scratch(size_t n, size_t lanes) {
    size_t tot = 0;
    for (size_t n_seg = (n + SEG - 1) / SEG; n_seg > 1; n_seg = (n_seg + SEG - 1) / SEG) tot += 2 * lanes * n_seg * 32;
    return tot;
}
static int suffix_horner(czk_ctx* ctx, const u64* in, size_t in_stride, size_t n, size_t lanes, Fr x, u64* out, size_t out_stride, unsigned shift,
                         u64* rem, char*& ws) {
    const size_t n_seg = (n + SEG - 1) / SEG;
    u64* carry = nullptr;
    if (n_seg > 1) {
        u64* H = (u64*)ws;
        ws += lanes * n_seg * 32;
        u64* R = (u64*)ws;
        ws += lanes * n_seg * 32;
        hipLaunchKernelGGL(k_seg_horner, dim3((unsigned)((n_seg + 255) / 256), (unsigned)lanes), dim3(256), 0, ctx->stream, in, in_stride, n, x, H, n_seg);
        CZK_TRY(suffix_horner(ctx, H, n_seg, n_seg, lanes, host_pow(x, SEG), R, n_seg, 0, nullptr, ws));
        carry = R;
    }
    hipLaunchKernelGGL(k_seg_fill, dim3((unsigned)((n_seg + 255) / 256), (unsigned)lanes), dim3(256), 0, ctx->stream, in, in_stride, n, x, carry, n_seg, out,
                       out_stride, shift, rem);
    CZK_HIP(ctx, hipGetLastError());
    return CZK_OK;
}

// p(x) alone (no quotient): the per-segment Horner values are the coefficients of a polynomial in x^SEG, SEG times shorter; the last
// level (one segment) writes the value.  Reads the input once and writes n / SEG elements, where the division writes n.
static size_t eval_horner_scratch(size_t n, size_t lanes) {
    size_t tot = 0;
    for (size_t n_seg = (n + SEG - 1) / SEG; n_seg > 1; n_seg = (n_seg + SEG - 1) / SEG) tot += lanes * n_seg * 32;
    return tot;
}
static int eval_horner(czk_ctx* ctx, const u64* in, size_t in_stride, size_t n, size_t lanes, Fr x, u64* value, char*& ws) {
    const size_t n_seg = (n + SEG - 1) / SEG;
    u64* H = value;   // one segment: its Horner value is p(x) (lanes x 1 elements: the layout of `value`)
    if (n_seg > 1) {
        H = (u64*)ws;
        ws += lanes * n_seg * 32;
    }
    hipLaunchKernelGGL(k_seg_horner, dim3((unsigned)((n_seg + 255) / 256), (unsigned)lanes), dim3(256), 0, ctx->stream, in, in_stride, n, x, H, n_seg);
    CZK_HIP(ctx, hipGetLastError());
    if (n_seg > 1) return eval_horner(ctx, H, n_seg, n_seg, lanes, host_pow(x, SEG), value, ws);
    return CZK_OK;
}

// Several polynomials in one launch per level (czk_poly_evaluate_many): the descriptors travel with the launch, blockIdx.z selects the polynomial.  A
// prover's evaluation round is two dozen evaluations whose upper levels are a few dozen elements each -- launch-bound when issued one by one.
constexpr unsigned EVAL_MANY = 16;
struct EvalDesc {
    const u64* in;
    size_t in_stride, n, n_seg;
    u64* out;        // lanes x n_seg segment values (the next level's input), or the lanes values when n_seg == 1
    Fr x;
    unsigned lanes;
};
struct EvalBatch {
    EvalDesc d[EVAL_MANY];
};
__global__ __launch_bounds__(256) void k_seg_horner_many(EvalBatch b) {
    const EvalDesc& d = b.d[blockIdx.z];
    const unsigned lane = blockIdx.y;
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (lane >= d.lanes || t >= d.n_seg) return;
    const u64* p = d.in + 4 * (size_t)lane * d.in_stride;
    const size_t a = t * SEG, e = a + SEG < d.n ? a + SEG : d.n;
    Fr r = Fr::zero();
    for (size_t i = e; i-- > a;) r = fp_add(fp_mul(r, d.x), pfr_load(p, i));
    pfr_store(d.out + 4 * (size_t)lane * d.n_seg, t, r);
}

// out[l][i] = sum_k c_k a_k[l][i] over terms of different lengths (czk_fr_lincomb): one read of every operand, one write of the result, where a chain of
// scale / resize / add calls makes a pass per call.  A term on one lane is PUBLIC: it is added on the lanes of `lift_mask` only (the reference's
// `shift` of a shared value by a public one: the king's lanes under SPDZ, every lane under GSZ).
constexpr unsigned LINCOMB_MAX = 12;
struct LinTerm {
    const u64* a;
    size_t len, lane_stride;   // lane_stride == 0: a public term
    Fr c;
    unsigned unit;             // c == 1: no multiplication
};
struct LinBatch {
    LinTerm t[LINCOMB_MAX];
    unsigned count, has_const;
    unsigned long long lift_mask;
    Fr constant;               // a PUBLIC constant on every element (lifting lanes only, like a public term)
};
__global__ __launch_bounds__(256) void k_lincomb(LinBatch b, u64* out, size_t out_len) {
    const unsigned lane = blockIdx.y;
    const bool lifts = (b.lift_mask >> (lane & 63)) & 1;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < out_len; i += (size_t)gridDim.x * blockDim.x) {
        Fr acc = (b.has_const && lifts) ? b.constant : Fr::zero();
        for (unsigned k = 0; k < b.count; k++) {
            const LinTerm& t = b.t[k];
            if (i >= t.len || (t.lane_stride == 0 && !lifts)) continue;
            const Fr v = pfr_load(t.a + 4 * (size_t)lane * t.lane_stride, i);
            acc = fp_add(acc, t.unit ? v : fp_mul(v, t.c));
        }
        pfr_store(out + 4 * (size_t)lane * out_len, i, acc);
    }
}

// ------------------------------------------------------------------------------------------------
// Running products out[i] = x_0 * ... * x_i of a public vector: the local loop of partial_products between its open and
// the final scale (mpc-algebra/src/share/field.rs:169-172; Plonk's grand product).  Same three-phase segment scheme as
// the suffix Horner sums, with multiplication as the operator.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_seg_product(const u64* in, size_t n, u64* S, size_t n_seg) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_seg) return;
    size_t a = t * SEG, b = a + SEG < n ? a + SEG : n;
    Fr r = pfr_load(in, a);
    for (size_t i = a + 1; i < b; i++) r = fp_mul(r, pfr_load(in, i));
    pfr_store(S, t, r);
}
// carry[t - 1] = product of everything before segment t (inclusive scan of the segment products)
__global__ __launch_bounds__(256) void k_seg_product_fill(const u64* in, size_t n, const u64* carry, size_t n_seg, u64* out) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_seg) return;
    size_t a = t * SEG, b = a + SEG < n ? a + SEG : n;
    Fr r = pfr_load(in, a);
    if (carry && t) r = fp_mul(pfr_load(carry, t - 1), r);
    pfr_store(out, a, r);
    for (size_t i = a + 1; i < b; i++) {
        r = fp_mul(r, pfr_load(in, i));
        pfr_store(out, i, r);
    }
}
static int prefix_product(czk_ctx* ctx, const u64* in, size_t n, u64* out, char*& ws) {
    const size_t n_seg = (n + SEG - 1) / SEG;
    u64* carry = nullptr;
    if (n_seg > 1) {
        u64* S = (u64*)ws;
        ws += n_seg * 32;
        u64* C = (u64*)ws;
        ws += n_seg * 32;
        hipLaunchKernelGGL(k_seg_product, dim3((unsigned)((n_seg + 255) / 256)), dim3(256), 0, ctx->stream, in, n, S, n_seg);
        CZK_TRY(prefix_product(ctx, S, n_seg, C, ws));
        carry = C;
    }
    hipLaunchKernelGGL(k_seg_product_fill, dim3((unsigned)((n_seg + 255) / 256)), dim3(256), 0, ctx->stream, in, n, carry, n_seg, out);
    CZK_HIP(ctx, hipGetLastError());
    return CZK_OK;
}

// ------------------------------------------------------------------------------------------------
// batch_inversion_and_mul (algebra/ff/src/fields/mod.rs:616-677): out[i] = coeff / v[i], zeros stay zero.  Montgomery's
// trick per 64-element segment (the reference's `parallel` build chunks the vector the same way, :632-640): running
// products into `out`, one Fermat inversion per segment, backward sweep.  Field inverses are unique, so the limbs equal
// the reference's whatever the chunking.
// ------------------------------------------------------------------------------------------------
constexpr unsigned INV_SEG = 64;
__global__ __launch_bounds__(128) void k_batch_inverse(const u64* v, size_t n, Fr coeff, u64* out) {
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t a = t * INV_SEG;
    if (a >= n) return;
    size_t b = a + INV_SEG < n ? a + INV_SEG : n;
    Fr run = Fr::one();
    for (size_t i = a; i < b; i++) {
        Fr x = pfr_load(v, i);
        if (!x.is_zero()) run = fp_mul(run, x);
        pfr_store(out, i, run);
    }
    Fr tmp = fp_mul(fp_inv(run), coeff);   // run != 0: a product of nonzero elements (or one)
    for (size_t i = b; i-- > a;) {
        Fr x = pfr_load(v, i);
        if (x.is_zero()) {
            pfr_store(out, i, Fr::zero());
            continue;
        }
        Fr s = i > a ? pfr_load(out, i - 1) : Fr::one();
        pfr_store(out, i, fp_mul(tmp, s));
        tmp = fp_mul(tmp, x);
    }
}

}  // namespace czk

using namespace czk;

struct czk_r1cs_matrix {
    int device = 0;   // GPU ordinal (the handle may outlive its context)
    size_t m = 0, nnz = 0, n_vars = 0;
    u32 *row_ptr = nullptr, *col = nullptr;
    u64* coeff = nullptr;
};

extern "C" void czk_r1cs_matrix_release(czk_r1cs_matrix* a) {
    if (!a) return;
    (void)hipSetDevice(a->device);
    if (a->row_ptr) (void)hipFree(a->row_ptr);
    if (a->col) (void)hipFree(a->col);
    if (a->coeff) (void)hipFree(a->coeff);
    delete a;
}

extern "C" int czk_r1cs_matrix_register(czk_ctx* ctx, const uint64_t* row_ptr, const uint32_t* col_idx, const uint64_t* coeff, size_t m, size_t nnz,
                                        size_t n_vars, int mem, czk_r1cs_matrix** out) {
    if (!ctx || !out || !row_ptr || (nnz && (!col_idx || !coeff))) return ctx ? set_err(ctx, CZK_ERR_ARG, "null r1cs matrix argument") : CZK_ERR_ARG;
    *out = nullptr;
    if (!valid_mem(mem)) return set_err(ctx, CZK_ERR_ARG, "mem must be CZK_MEM_HOST or CZK_MEM_DEVICE");
    if (nnz >= ((size_t)1 << 32) || n_vars >= ((size_t)1 << 31) || m >= ((size_t)1 << 32))
        return set_err(ctx, CZK_ERR_SIZE, "r1cs matrix too large for 32-bit indices");
    CZK_HIP(ctx, hipSetDevice(ctx->device));
    czk_r1cs_matrix* a = new czk_r1cs_matrix();
    a->device = ctx->device;
    a->m = m;
    a->nnz = nnz;
    a->n_vars = n_vars;
    u64* rp64 = nullptr;
    u32* bad = nullptr;
    int rc = CZK_OK;
    auto fail = [&](int code, const std::string& msg) {
        rc = set_err(ctx, code, msg);
    };
    const hipMemcpyKind kind = mem == CZK_MEM_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
    if (hipMalloc(&a->row_ptr, (m + 1) * 4) != hipSuccess || hipMalloc(&a->col, (nnz ? nnz : 1) * 4) != hipSuccess ||
        hipMalloc(&a->coeff, (nnz ? nnz : 1) * 32) != hipSuccess || hipMalloc(&rp64, (m + 1) * 8) != hipSuccess || hipMalloc(&bad, 4) != hipSuccess)
        fail(CZK_ERR_NOMEM, "hipMalloc r1cs matrix");
    if (rc == CZK_OK && (hipMemcpyAsync(rp64, row_ptr, (m + 1) * 8, kind, ctx->stream) != hipSuccess ||
                         (nnz && hipMemcpyAsync(a->col, col_idx, nnz * 4, kind, ctx->stream) != hipSuccess) ||
                         (nnz && hipMemcpyAsync(a->coeff, coeff, nnz * 32, kind, ctx->stream) != hipSuccess) ||
                         hipMemsetAsync(bad, 0, 4, ctx->stream) != hipSuccess))
        fail(CZK_ERR_HIP, "copy r1cs matrix");
    if (rc == CZK_OK) {
        size_t work = nnz > m + 1 ? nnz : m + 1;
        hipLaunchKernelGGL(k_r1cs_prepare, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, ctx->stream, rp64, a->col, a->coeff, m, nnz, n_vars,
                           a->row_ptr, bad);
        u32 hb = 0;
        if (hipMemcpyAsync(&hb, bad, 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess || hipStreamSynchronize(ctx->stream) != hipSuccess)
            fail(CZK_ERR_HIP, "r1cs matrix validation");
        else if (hb & 1) fail(CZK_ERR_ARG, "r1cs matrix: row_ptr is not a monotone CSR offset array ending at nnz");
        else if (hb & 2) fail(CZK_ERR_ARG, "r1cs matrix: variable index out of range (the reference would panic on assignment[index])");
    }
    if (rp64) (void)hipFree(rp64);
    if (bad) (void)hipFree(bad);
    if (rc != CZK_OK) {
        czk_r1cs_matrix_release(a);
        return rc;
    }
    *out = a;
    return CZK_OK;
}

extern "C" int czk_r1cs_matvec(czk_ctx* ctx, const czk_r1cs_matrix* a, const uint64_t* z, size_t z_stride, size_t lanes, uint64_t* out,
                               size_t out_stride, int mem) {
    if (!ctx || !a || (lanes && (!z || !out))) return ctx ? set_err(ctx, CZK_ERR_ARG, "null r1cs_matvec argument") : CZK_ERR_ARG;
    if (!valid_mem(mem)) return set_err(ctx, CZK_ERR_ARG, "mem must be CZK_MEM_HOST or CZK_MEM_DEVICE");
    if (z_stride < a->n_vars || out_stride < a->m) return set_err(ctx, CZK_ERR_SIZE, "r1cs_matvec: assignment shorter than n_vars or output shorter than m");
    if (!lanes || !a->m) return CZK_OK;
    CZK_HIP(ctx, hipSetDevice(ctx->device));
    Staged sz{ctx}, so{ctx};
    CZK_TRY(sz.to_device(z, lanes * z_stride * 32, mem));
    CZK_TRY(so.to_device(mem == CZK_MEM_HOST ? nullptr : out, lanes * out_stride * 32, mem));
    if (so.owned) CZK_HIP(ctx, hipMemsetAsync(so.dev, 0, lanes * out_stride * 32, ctx->stream));
    size_t blocks = (a->m + 255) / 256, cap = (size_t)ctx->num_cu * 16;
    if (blocks > cap) blocks = cap;
    {
        ProfScope ps(ctx, "r1cs_matvec");
        hipLaunchKernelGGL(k_r1cs_matvec, dim3((unsigned)blocks, (unsigned)lanes), dim3(256), 0, ctx->stream, a->row_ptr, a->col, a->coeff, a->m,
                           (const u64*)sz.dev, z_stride, (u64*)so.dev, out_stride);
    }
    CZK_HIP(ctx, hipGetLastError());
    if (so.owned) {
        // only rows [0, m) of each lane are defined by this call: copy those back
        for (size_t l = 0; l < lanes; l++)
            CZK_HIP(ctx, hipMemcpyAsync(out + 4 * l * out_stride, (const u64*)so.dev + 4 * l * out_stride, a->m * 32, hipMemcpyDeviceToHost, ctx->stream));
        CZK_HIP(ctx, hipStreamSynchronize(ctx->stream));
    }
    return CZK_OK;
}

extern "C" int czk_poly_div_linear(czk_ctx* ctx, const uint64_t* coeffs, size_t n, size_t lanes, const uint64_t* z, uint64_t* quotient,
                                   uint64_t* remainder, int mem) {
    if (!ctx || !z || (lanes && n && !coeffs) || (lanes && n > 1 && !quotient)) return ctx ? set_err(ctx, CZK_ERR_ARG, "null poly_div argument") : CZK_ERR_ARG;
    if (!valid_mem(mem)) return set_err(ctx, CZK_ERR_ARG, "mem must be CZK_MEM_HOST or CZK_MEM_DEVICE");
    if (!lanes) return CZK_OK;
    CZK_HIP(ctx, hipSetDevice(ctx->device));
    if (n == 0) {   // zero polynomial: zero quotient, zero remainder
        if (remainder) {
            if (mem == CZK_MEM_HOST) memset(remainder, 0, lanes * 32);
            else CZK_HIP(ctx, hipMemsetAsync(remainder, 0, lanes * 32, ctx->stream));
        }
        return CZK_OK;
    }
    const Fr x = host_fr(z);   // host pointer, 8-byte aligned only
    const size_t qn = n - 1;
    Staged sp{ctx}, sq{ctx}, sr{ctx};
    CZK_TRY(sp.to_device(coeffs, lanes * n * 32, mem));
    CZK_TRY(sq.to_device(mem == CZK_MEM_HOST ? nullptr : quotient, lanes * qn * 32, mem));
    CZK_TRY(sr.to_device(mem == CZK_MEM_HOST ? nullptr : remainder, lanes * 32, remainder ? mem : CZK_MEM_HOST));
    CZK_TRY(ensure_buf(ctx, ctx->poly_scratch, suffix_horner_scratch(n, lanes) + 256));
    char* ws = (char*)ctx->poly_scratch.p;
    {
        ProfScope ps(ctx, "poly_div_linear");
        CZK_TRY(suffix_horner(ctx, (const u64*)sp.dev, n, n, lanes, x, (u64*)sq.dev, qn, 1, (u64*)sr.dev, ws));
    }
    CZK_TRY(sq.to_host(quotient, lanes * qn * 32));
    if (remainder) CZK_TRY(sr.to_host(remainder, lanes * 32));
    return CZK_OK;
}

extern "C" int czk_poly_div_vanishing(czk_ctx* ctx, const uint64_t* coeffs, size_t m, size_t lanes, size_t n, uint64_t* quotient, uint64_t* remainder, int mem) {
    if (!ctx || !n || (lanes && m && !coeffs) || (lanes && m > n && !quotient)) return ctx ? set_err(ctx, CZK_ERR_ARG, "null / zero poly_div_vanishing argument") : CZK_ERR_ARG;
    if (!valid_mem(mem)) return set_err(ctx, CZK_ERR_ARG, "mem must be CZK_MEM_HOST or CZK_MEM_DEVICE");
    if (!lanes) return CZK_OK;
    if (m < n) return set_err(ctx, CZK_ERR_ARG, "czk_poly_div_vanishing: fewer than n coefficients (pad with zeros: the remainder is the polynomial itself)");
    if (lanes > 65535) return set_err(ctx, CZK_ERR_SIZE, "too many lanes");
    CZK_HIP(ctx, hipSetDevice(ctx->device));
    Staged sp{ctx}, sq{ctx}, sr{ctx};
    CZK_TRY(sp.to_device(coeffs, lanes * m * 32, mem));
    CZK_TRY(sq.to_device(mem == CZK_MEM_HOST ? nullptr : quotient, lanes * (m - n) * 32, mem));
    CZK_TRY(sr.to_device(mem == CZK_MEM_HOST ? nullptr : remainder, lanes * n * 32, remainder ? mem : CZK_MEM_HOST));
    {
        ProfScope ps(ctx, "poly_div_vanishing");
        hipLaunchKernelGGL(k_div_vanishing, dim3((unsigned)((n + 255) / 256), (unsigned)lanes), dim3(256), 0, ctx->stream, (const u64*)sp.dev, m, n, (u64*)sq.dev,
                           remainder ? (u64*)sr.dev : nullptr);
        CZK_HIP(ctx, hipGetLastError());
    }
    CZK_TRY(sq.to_host(quotient, lanes * (m - n) * 32));
    if (remainder) CZK_TRY(sr.to_host(remainder, lanes * n * 32));
    return CZK_OK;
}

extern "C" int czk_poly_evaluate(czk_ctx* ctx, const uint64_t* coeffs, size_t n, size_t lanes, const uint64_t* z, uint64_t* values, int mem) {
    if (!ctx || !z || (lanes && n && !coeffs) || (lanes && !values)) return ctx ? set_err(ctx, CZK_ERR_ARG, "null poly_evaluate argument") : CZK_ERR_ARG;
    if (!valid_mem(mem)) return set_err(ctx, CZK_ERR_ARG, "mem must be CZK_MEM_HOST or CZK_MEM_DEVICE");
    if (!lanes) return CZK_OK;
    CZK_HIP(ctx, hipSetDevice(ctx->device));
    if (n == 0) {   // the zero polynomial
        if (mem == CZK_MEM_HOST) memset(values, 0, lanes * 32);
        else CZK_HIP(ctx, hipMemsetAsync(values, 0, lanes * 32, ctx->stream));
        return CZK_OK;
    }
    const Fr x = host_fr(z);   // host pointer, 8-byte aligned only
    Staged sp{ctx}, sv{ctx};
    CZK_TRY(sp.to_device(coeffs, lanes * n * 32, mem));
    CZK_TRY(sv.to_device(mem == CZK_MEM_HOST ? nullptr : values, lanes * 32, mem));
    CZK_TRY(ensure_buf(ctx, ctx->poly_scratch, eval_horner_scratch(n, lanes) + 256));
    char* ws = (char*)ctx->poly_scratch.p;
    {
        ProfScope ps(ctx, "poly_evaluate");
        CZK_TRY(eval_horner(ctx, (const u64*)sp.dev, n, n, lanes, x, (u64*)sv.dev, ws));
    }
    CZK_TRY(sv.to_host(values, lanes * 32));
    return CZK_OK;
}

extern "C" int czk_poly_evaluate_many(czk_ctx* ctx, size_t count, const uint64_t* const* coeffs, const size_t* n, const size_t* lanes, const uint64_t* z,
                                      uint64_t* const* values) {
    if (!ctx || (count && (!coeffs || !n || !lanes || !z || !values))) return ctx ? set_err(ctx, CZK_ERR_ARG, "null poly_evaluate_many argument") : CZK_ERR_ARG;
    CZK_HIP(ctx, hipSetDevice(ctx->device));
    for (size_t at = 0; at < count; at += EVAL_MANY) {
        const unsigned m = (unsigned)(count - at < EVAL_MANY ? count - at : EVAL_MANY);
        // scratch: every polynomial keeps two level arrays of its own (lanes x n_seg, then 32 times smaller, ping-pong would do; the sum is n / 31 per lane)
        size_t need = 256;
        for (unsigned k = 0; k < m; k++) {
            if (lanes[at + k] > 65535) return set_err(ctx, CZK_ERR_SIZE, "too many lanes");
            if (lanes[at + k] && n[at + k] && (!coeffs[at + k] || !values[at + k])) return set_err(ctx, CZK_ERR_ARG, "null poly_evaluate_many polynomial / value pointer");
            need += eval_horner_scratch(n[at + k], lanes[at + k]) + 64;
        }
        CZK_TRY(ensure_buf(ctx, ctx->poly_scratch, need));
        char* ws = (char*)ctx->poly_scratch.p;
        struct Cur {
            const u64* in;
            size_t stride, n;
            Fr x;
            u64* value;
            unsigned lanes;
            bool done;
        } cur[EVAL_MANY];
        for (unsigned k = 0; k < m; k++) {
            cur[k] = Cur{(const u64*)coeffs[at + k], n[at + k], n[at + k], host_fr(z + 4 * (at + k)), (u64*)values[at + k], (unsigned)lanes[at + k], false};
            if (!cur[k].lanes) cur[k].done = true;
            else if (!cur[k].n) {   // the zero polynomial
                CZK_HIP(ctx, hipMemsetAsync(cur[k].value, 0, (size_t)cur[k].lanes * 32, ctx->stream));
                cur[k].done = true;
            }
        }
        ProfScope ps(ctx, "poly_evaluate");
        for (;;) {
            EvalBatch b;
            unsigned cnt = 0, max_lanes = 0;
            size_t max_seg = 0;
            unsigned which[EVAL_MANY];
            for (unsigned k = 0; k < m; k++) {
                if (cur[k].done) continue;
                EvalDesc& d = b.d[cnt];
                d.in = cur[k].in, d.in_stride = cur[k].stride, d.n = cur[k].n, d.n_seg = (cur[k].n + SEG - 1) / SEG, d.x = cur[k].x, d.lanes = cur[k].lanes;
                if (d.n_seg == 1) d.out = cur[k].value;
                else {
                    d.out = (u64*)ws;
                    ws += (size_t)d.lanes * d.n_seg * 32;
                }
                max_lanes = d.lanes > max_lanes ? d.lanes : max_lanes;
                max_seg = d.n_seg > max_seg ? d.n_seg : max_seg;
                which[cnt++] = k;
            }
            if (!cnt) break;
            hipLaunchKernelGGL(k_seg_horner_many, dim3((unsigned)((max_seg + 255) / 256), max_lanes, cnt), dim3(256), 0, ctx->stream, b);
            CZK_HIP(ctx, hipGetLastError());
            for (unsigned j = 0; j < cnt; j++) {
                Cur& c = cur[which[j]];
                const EvalDesc& d = b.d[j];
                if (d.n_seg == 1) c.done = true;
                else c.in = d.out, c.stride = d.n_seg, c.n = d.n_seg, c.x = host_pow(c.x, SEG);
            }
        }
    }
    return CZK_OK;
}

extern "C" int czk_fr_lincomb(czk_ctx* ctx, size_t count, const uint64_t* const* terms, const size_t* term_len, const size_t* term_lanes, const uint64_t* coeffs,
                              const uint64_t* constant, size_t lanes, uint64_t lift_mask, uint64_t* out, size_t out_len) {
    if (!ctx || (count && (!terms || !term_len || !term_lanes || !coeffs)) || (lanes && out_len && !out)) return ctx ? set_err(ctx, CZK_ERR_ARG, "null fr_lincomb argument") : CZK_ERR_ARG;
    if (count > LINCOMB_MAX) return set_err(ctx, CZK_ERR_SIZE, "czk_fr_lincomb: at most 12 terms per call (chain calls: the result of one is a unit term of the next)");
    if (lanes > 64) return set_err(ctx, CZK_ERR_SIZE, "czk_fr_lincomb: at most 64 lanes (the lift mask)");
    if (!lanes || !out_len) return CZK_OK;
    CZK_HIP(ctx, hipSetDevice(ctx->device));
    LinBatch b;
    b.count = (unsigned)count;
    b.lift_mask = lanes == 1 ? ~0ull : lift_mask;   // a public result takes every public term
    b.has_const = constant ? 1 : 0;
    b.constant = constant ? host_fr(constant) : Fr::zero();
    static const Fr ONE = Fr::one();
    for (size_t k = 0; k < count; k++) {
        if (term_lanes[k] != lanes && term_lanes[k] != 1) return set_err(ctx, CZK_ERR_ARG, "czk_fr_lincomb: a term has `lanes` lanes or one (public)");
        if (term_len[k] && !terms[k]) return set_err(ctx, CZK_ERR_ARG, "czk_fr_lincomb: null term");
        LinTerm& t = b.t[k];
        t.a = (const u64*)terms[k];
        t.len = term_len[k] < out_len ? term_len[k] : out_len;
        t.lane_stride = (term_lanes[k] == 1 && lanes > 1) ? 0 : term_len[k];
        t.c = host_fr(coeffs + 4 * k);
        t.unit = t.c == ONE ? 1 : 0;
    }
    ProfScope ps(ctx, "fr_lincomb");
    size_t blocks = (out_len + 255) / 256, cap = (size_t)ctx->num_cu * 16;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(k_lincomb, dim3((unsigned)blocks, (unsigned)lanes), dim3(256), 0, ctx->stream, b, (u64*)out, out_len);
    CZK_HIP(ctx, hipGetLastError());
    return CZK_OK;
}

extern "C" int czk_fr_prefix_product(czk_ctx* ctx, const uint64_t* x, size_t n, uint64_t* out, int mem) {
    if (!ctx || (n && (!x || !out))) return ctx ? set_err(ctx, CZK_ERR_ARG, "null prefix_product argument") : CZK_ERR_ARG;
    if (!valid_mem(mem)) return set_err(ctx, CZK_ERR_ARG, "mem must be CZK_MEM_HOST or CZK_MEM_DEVICE");
    if (!n) return CZK_OK;
    CZK_HIP(ctx, hipSetDevice(ctx->device));
    Staged sx{ctx}, so{ctx};
    CZK_TRY(sx.to_device(x, n * 32, mem));
    CZK_TRY(so.to_device(mem == CZK_MEM_HOST ? nullptr : out, n * 32, mem));
    CZK_TRY(ensure_buf(ctx, ctx->poly_scratch, suffix_horner_scratch(n, 1) + 256));
    char* ws = (char*)ctx->poly_scratch.p;
    {
        ProfScope ps(ctx, "fr_prefix_product");
        CZK_TRY(prefix_product(ctx, (const u64*)sx.dev, n, (u64*)so.dev, ws));
    }
    return so.to_host(out, n * 32);
}

extern "C" int czk_fr_batch_inverse(czk_ctx* ctx, const uint64_t* v, size_t n, const uint64_t* coeff, uint64_t* out, int mem) {
    if (!ctx || (n && (!v || !out))) return ctx ? set_err(ctx, CZK_ERR_ARG, "null batch_inverse argument") : CZK_ERR_ARG;
    if (!valid_mem(mem)) return set_err(ctx, CZK_ERR_ARG, "mem must be CZK_MEM_HOST or CZK_MEM_DEVICE");
    if (mem == CZK_MEM_DEVICE && v == out) return set_err(ctx, CZK_ERR_ARG, "batch_inverse: out must not alias v in device memory");
    if (!n) return CZK_OK;
    CZK_HIP(ctx, hipSetDevice(ctx->device));
    const Fr k = coeff ? host_fr(coeff) : Fr::one();   // host pointer, 8-byte aligned only
    Staged sv{ctx}, so{ctx};
    CZK_TRY(sv.to_device(v, n * 32, mem));
    CZK_TRY(so.to_device(mem == CZK_MEM_HOST ? nullptr : out, n * 32, mem));
    const size_t segs = (n + INV_SEG - 1) / INV_SEG;
    {
        ProfScope ps(ctx, "fr_batch_inverse");
        hipLaunchKernelGGL(k_batch_inverse, dim3((unsigned)((segs + 127) / 128)), dim3(128), 0, ctx->stream, (const u64*)sv.dev, n, k, (u64*)so.dev);
    }
    CZK_HIP(ctx, hipGetLastError());
    return so.to_host(out, n * 32);
}
