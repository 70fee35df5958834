/* czk_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the reference's hot path, limb-exact:
 * Fp256/Fp384 Montgomery arithmetic, Fp2, short-Weierstrass Jacobian group law, Pippenger
 * VariableBaseMSM with the reference window rule, the fffft-style in-order radix-2 FFT with the
 * reference's root selection (LARGE_SUBGROUP_ROOT_OF_UNITY^3) and coset shift (22), and the Groth16
 * witness-map / MSM sequence, the mixed-radix (3 * 2^k) FFT and the GSZ share / open arithmetic.  Single-threaded like the
 * reference's build, plus an OpenMP task-parallel variant of the same code (orc_groth16_local_par) used only by the all-core
 * `cpu_baseline` leg.  It is the checker for tests/, __graft_entry__.smoke() and the timed
 * `cpu_baseline` leg of bench.py; the product library (collaborative-zksnark_amd/csrc) never links,
 * loads or calls it.
 *
 * Pinning: the reference cannot be built here (Rust nightly + crates.io, no cargo in the image) and holds
 * no golden NTT/MSM vectors, so this restatement is pinned against (a) the reference's constant KATs, with every constant
 * compared against the reference's source text (tools/check_constants_vs_reference.py -> tests/golden/reference_constants.json),
 * and (b) the independent Python big-int oracle oracle/pyref.py, via tests/test_oracle_*.py and the
 * fixtures under tests/golden/.  Beyond those it is "parity unpinned" (see DESIGN.md).
 *
 * Paths in comments are relative to /root/reference.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <omp.h>

typedef unsigned __int128 u128;

/* utils/src/lib.rs:65-73 -- ark_std::log2 = ceil(log2(x)), log2(0) = 0 */
static unsigned orc_log2(size_t x) {
    if (x == 0) return 0;
    unsigned lz = (unsigned)__builtin_clzll((unsigned long long)x);
    if ((x & (x - 1)) == 0) return 63 - lz;
    return 64 - lz;
}

/* ------------------------------------------------------------------ Fr = Fp256<FrParameters> */
/* curves/bls12_377/src/fields/fr.rs:30-75 */
static const uint64_t fr_MODULUS[4] = {725501752471715841ULL, 6461107452199829505ULL, 6968279316240510977ULL,
                                       1345280370688173398ULL};
static const uint64_t fr_R[4] = {9015221291577245683ULL, 8239323489949974514ULL, 1646089257421115374ULL,
                                 958099254763297437ULL};
static const uint64_t fr_R2[4] = {2726216793283724667ULL, 14712177743343147295ULL, 12091039717619697043ULL,
                                  81024008013859129ULL};
#define fr_INV 725501752471715839ULL
/* fr.rs:69-74 GENERATOR (Montgomery limbs; decodes to 22) */
static const uint64_t fr_GENERATOR[4] = {2984901390528151251ULL, 10561528701063790279ULL, 5476750214495080041ULL,
                                         898978044469942640ULL};
/* fr.rs:21-28 LARGE_SUBGROUP_ROOT_OF_UNITY (Montgomery limbs), SMALL_SUBGROUP_BASE = 3, adicity 1 */
static const uint64_t fr_LARGE_ROOT[4] = {0x9bfe9d90c790c167ULL, 0x7175a69e39013bffULL, 0x3fbbb698adabcf93ULL,
                                          0xc59f8d8d6f0dc97ULL};
#define FR_TWO_ADICITY 47

#define FP_N 4
#define FP(x) fr_##x
#define FP_T fr_t
#include "fp_tmpl.h"
#undef FP_N
#undef FP
#undef FP_T

/* ------------------------------------------------------------------ Fq = Fp384<FqParameters> */
/* curves/bls12_377/src/fields/fq.rs:23-62 */
static const uint64_t fq_MODULUS[6] = {0x8508c00000000001ULL, 0x170b5d4430000000ULL, 0x1ef3622fba094800ULL,
                                       0x1a22d9f300f5138fULL, 0xc63b05c06ca1493bULL, 0x1ae3a4617c510eaULL};
static const uint64_t fq_R[6] = {202099033278250856ULL, 5854854902718660529ULL, 11492539364873682930ULL,
                                 8885205928937022213ULL, 5545221690922665192ULL, 39800542322357402ULL};
static const uint64_t fq_R2[6] = {0xb786686c9400cd22ULL, 0x329fcaab00431b1ULL, 0x22a5f11162d6b46dULL,
                                  0xbfdf7d03827dc3acULL, 0x837e92f041790bf9ULL, 0x6dfccb1e914b88ULL};
#define fq_INV 9586122913090633727ULL

#define FP_N 6
#define FP(x) fq_##x
#define FP_T fq_t
#include "fp_tmpl.h"
#undef FP_N
#undef FP
#undef FP_T

/* ------------------------------------------------------------------ Fq2 = Fq[u]/(u^2 + 5) */
/* algebra/ff/src/fields/models/quadratic_extension.rs; NONRESIDUE = -5 (curves/bls12_377/src/fields/fq2.rs:13) */
typedef struct { fq_t c0, c1; } fq2_t;

/* fq2.rs:29-34 -- mul_fp_by_nonresidue: -(2x) doubled, minus x  =  -5x */
static void fq_mul_by_nonresidue(fq_t *r, const fq_t *x) {
    fq_t t;
    fq_dbl(&t, x);
    fq_neg(&t, &t);
    fq_dbl(&t, &t);
    fq_sub(r, &t, x);
}
static void fq2_zero(fq2_t *a) { fq_zero(&a->c0); fq_zero(&a->c1); }
static void fq2_one(fq2_t *a) { fq_one(&a->c0); fq_zero(&a->c1); }
static int fq2_is_zero(const fq2_t *a) { return fq_is_zero(&a->c0) && fq_is_zero(&a->c1); }
static int fq2_is_one(const fq2_t *a) { return fq_is_one(&a->c0) && fq_is_zero(&a->c1); }
static int fq2_eq(const fq2_t *a, const fq2_t *b) { return fq_eq(&a->c0, &b->c0) && fq_eq(&a->c1, &b->c1); }
/* quadratic_extension.rs:552-565 */
static void fq2_add(fq2_t *r, const fq2_t *a, const fq2_t *b) { fq_add(&r->c0, &a->c0, &b->c0); fq_add(&r->c1, &a->c1, &b->c1); }
static void fq2_sub(fq2_t *r, const fq2_t *a, const fq2_t *b) { fq_sub(&r->c0, &a->c0, &b->c0); fq_sub(&r->c1, &a->c1, &b->c1); }
/* :227-231 */
static void fq2_dbl(fq2_t *r, const fq2_t *a) { fq_dbl(&r->c0, &a->c0); fq_dbl(&r->c1, &a->c1); }
static void fq2_neg(fq2_t *r, const fq2_t *a) { fq_neg(&r->c0, &a->c0); fq_neg(&r->c1, &a->c1); }
/* :571-583 -- Karatsuba */
static void fq2_mul(fq2_t *r, const fq2_t *a, const fq2_t *b) {
    fq_t v0, v1, s, t, nr;
    fq_mul(&v0, &a->c0, &b->c0);
    fq_mul(&v1, &a->c1, &b->c1);
    fq_add(&s, &a->c1, &a->c0);
    fq_add(&t, &b->c0, &b->c1);
    fq_mul(&s, &s, &t);
    fq_sub(&s, &s, &v0);
    fq_sub(&s, &s, &v1);
    fq_mul_by_nonresidue(&nr, &v1);          /* add_and_mul_base_field_by_nonresidue(v0, v1) = v0 + beta*v1 */
    fq_add(&r->c0, &v0, &nr);
    r->c1 = s;
}
/* :257-305 -- square_in_place, generic (beta != -1) branch */
static void fq2_sqr(fq2_t *r, const fq2_t *a) {
    fq_t v0, v3, v2, nr, t;
    fq_sub(&v0, &a->c0, &a->c1);
    fq_mul_by_nonresidue(&nr, &a->c1);
    fq_sub(&v3, &a->c0, &nr);                /* sub_and_mul_base_field_by_nonresidue: c0 - beta*c1 */
    fq_mul(&v2, &a->c0, &a->c1);
    fq_mul(&v0, &v0, &v3);
    fq_dbl(&r->c1, &v2);
    /* add_and_mul_base_field_by_nonresidue_plus_one(v0, v2) = (v0 + v2) + beta*v2  (quadratic_extension.rs:71-78) */
    fq_add(&t, &v0, &v2);
    fq_mul_by_nonresidue(&nr, &v2);
    fq_add(&r->c0, &t, &nr);
}
/* :308-324 */
static int fq2_inv(fq2_t *r, const fq2_t *a) {
    if (fq2_is_zero(a)) return 0;
    fq_t v1, v0, nr, t;
    fq_sqr(&v1, &a->c1);
    fq_sqr(&t, &a->c0);
    fq_mul_by_nonresidue(&nr, &v1);
    fq_sub(&v0, &t, &nr);
    fq_inv(&v1, &v0);
    fq_mul(&r->c0, &a->c0, &v1);
    fq_mul(&t, &a->c1, &v1);
    fq_neg(&r->c1, &t);
    return 1;
}

/* ------------------------------------------------------------------ G1 / G2 */
#define BF(x) fq_##x
#define BF_T fq_t
#define EC(x) g1_##x
#include "ec_tmpl.h"
#undef BF
#undef BF_T
#undef EC

#define BF(x) fq2_##x
#define BF_T fq2_t
#define EC(x) g2_##x
#include "ec_tmpl.h"
#undef BF
#undef BF_T
#undef EC

/* ------------------------------------------------------------------ Radix2EvaluationDomain<Fr> */
typedef struct {
    uint64_t size;
    unsigned log_size;
    fr_t size_inv, group_gen, group_gen_inv, generator, generator_inv;
} orc_domain_t;

/* algebra/ff/src/fields/mod.rs:337-386 -- get_root_of_unity, LARGE_SUBGROUP branch (n a power of two:
 * q_adicity = 0 so omega = LARGE^3, then squared TWO_ADICITY - log2(n) times). */
static int orc_root_of_unity(fr_t *omega, unsigned log_n) {
    if (log_n > FR_TWO_ADICITY) return 0;
    fr_t w;
    memcpy(w.l, fr_LARGE_ROOT, sizeof w.l);
    const uint64_t three[1] = {3};
    fr_pow(&w, &w, three, 1);
    for (unsigned i = log_n; i < FR_TWO_ADICITY; i++) fr_sqr(&w, &w);
    *omega = w;
    return 1;
}

/* algebra/poly/src/domain/radix2/mod.rs:51-82 -- Radix2EvaluationDomain::new */
static int orc_domain_new(orc_domain_t *d, size_t num_coeffs) {
    uint64_t size = 1;
    unsigned lg = 0;
    while (size < num_coeffs) { size <<= 1; lg++; }
    if (lg > FR_TWO_ADICITY) return 0;
    d->size = size;
    d->log_size = lg;
    if (!orc_root_of_unity(&d->group_gen, lg)) return 0;
    fr_t sz;
    fr_from_u64(&sz, size);
    fr_inv(&d->size_inv, &sz);
    fr_inv(&d->group_gen_inv, &d->group_gen);
    memcpy(d->generator.l, fr_GENERATOR, sizeof d->generator.l);
    fr_inv(&d->generator_inv, &d->generator);
    return 1;
}

/* radix2/fft.rs:248-260 -- derange (bit-reversal permutation) */
static void orc_derange(fr_t *x, size_t n, unsigned log_n) {
    for (uint64_t idx = 1; idx + 1 < n; idx++) {
        uint64_t r = 0, t = idx;
        for (unsigned b = 0; b < log_n; b++) { r = (r << 1) | (t & 1); t >>= 1; }
        if (idx < r) { fr_t tmp = x[idx]; x[idx] = x[r]; x[r] = tmp; }
    }
}
/* domain/utils.rs:22-39 + radix2/fft.rs:75-78 -- roots_of_unity: size/2 successive powers */
static fr_t *orc_roots(size_t half, const fr_t *root) {
    fr_t *roots = (fr_t *)malloc((half ? half : 1) * sizeof(fr_t));
    fr_t v;
    fr_one(&v);
    for (size_t i = 0; i < half; i++) { roots[i] = v; fr_mul(&v, &v, root); }
    return roots;
}
/* radix2/fft.rs:140-203 -- io_helper (DIF, gap n/2 .. 1), serial build incl. root compaction :194-200 */
static void orc_io_helper(fr_t *x, size_t n, const fr_t *root) {
    fr_t *roots = orc_roots(n / 2, root);
    size_t root_len = n / 2;
    for (size_t gap = n / 2; gap > 0; gap /= 2) {
        for (size_t base = 0; base < n; base += 2 * gap) {
            for (size_t k = 0; k < gap; k++) {
                fr_t *lo = &x[base + k], *hi = &x[base + gap + k], neg;
                fr_sub(&neg, lo, hi);
                fr_add(lo, lo, hi);
                fr_mul(hi, &neg, &roots[k]);
            }
        }
        for (size_t i = 1; i < root_len / 2; i++) roots[i] = roots[2 * i];
        root_len /= 2;
    }
    free(roots);
}
/* radix2/fft.rs:205-235 -- oi_helper (DIT, gap 1 .. n/2) */
static void orc_oi_helper(fr_t *x, size_t n, const fr_t *root) {
    fr_t *roots = orc_roots(n / 2, root);
    for (size_t gap = 1; gap < n; gap *= 2) {
        size_t nchunks = n / (2 * gap);
        for (size_t base = 0; base < n; base += 2 * gap) {
            for (size_t k = 0; k < gap; k++) {
                fr_t *lo = &x[base + k], *hi = &x[base + gap + k], neg;
                fr_mul(hi, hi, &roots[nchunks * k]);
                fr_sub(&neg, lo, hi);
                fr_add(lo, lo, hi);
                *hi = neg;
            }
        }
    }
    free(roots);
}
/* domain/mod.rs:99-106 -- distribute_powers_and_mul_by_const (serial): x[i] *= c * g^i */
static void orc_distribute_powers(fr_t *x, size_t n, const fr_t *g, const fr_t *c) {
    fr_t pw = *c;
    for (size_t i = 0; i < n; i++) { fr_mul(&x[i], &x[i], &pw); fr_mul(&pw, &pw, g); }
}

enum { ORC_FFT = 0, ORC_IFFT = 1, ORC_COSET_FFT = 2, ORC_COSET_IFFT = 3 };

/* {fft, ifft, coset_fft, coset_ifft}_in_place on a buffer of D = 2^log_d elements whose first in_len
 * entries are the caller's vector (the tail is overwritten with zero = `resize(size, T::zero())`).
 * radix2/mod.rs:99-117, domain/mod.rs:139-142, radix2/fft.rs:22-35. */
int orc_ntt_fr(uint64_t *data, unsigned log_d, int kind, size_t in_len) {
    orc_domain_t d;
    size_t n = (size_t)1 << log_d;
    if (!orc_domain_new(&d, n) || in_len > n) return 1;
    fr_t *x = (fr_t *)data;
    fr_t one;
    fr_one(&one);
    if (kind == ORC_COSET_FFT) orc_distribute_powers(x, in_len, &d.generator, &one);  /* on the un-resized input */
    for (size_t i = in_len; i < n; i++) fr_zero(&x[i]);
    if (kind == ORC_FFT || kind == ORC_COSET_FFT) {
        orc_io_helper(x, n, &d.group_gen);
        orc_derange(x, n, log_d);
    } else {
        orc_derange(x, n, log_d);
        orc_oi_helper(x, n, &d.group_gen_inv);
        if (kind == ORC_IFFT) {
            for (size_t i = 0; i < n; i++) fr_mul(&x[i], &x[i], &d.size_inv);
        } else {
            orc_distribute_powers(x, n, &d.generator_inv, &d.size_inv);
        }
    }
    return 0;
}

/* Domain constants, for cross-checking the product's host-side domain object. out = 6 x 4 limbs:
 * size_inv, group_gen, group_gen_inv, generator, generator_inv, vanishing_on_coset_inv (g^D - 1)^-1 */
int orc_domain_constants(unsigned log_d, uint64_t *out) {
    orc_domain_t d;
    if (!orc_domain_new(&d, (size_t)1 << log_d)) return 1;
    fr_t v, one;
    uint64_t e[1] = {(uint64_t)1 << log_d};
    fr_pow(&v, &d.generator, e, 1);        /* evaluate_vanishing_polynomial(g) = g^D - 1 (radix2/mod.rs) */
    fr_one(&one);
    fr_sub(&v, &v, &one);
    fr_inv(&v, &v);
    memcpy(out + 0, d.size_inv.l, 32);
    memcpy(out + 4, d.group_gen.l, 32);
    memcpy(out + 8, d.group_gen_inv.l, 32);
    memcpy(out + 12, d.generator.l, 32);
    memcpy(out + 16, d.generator_inv.l, 32);
    memcpy(out + 20, v.l, 32);
    return 0;
}

/* Polynomial evaluation by Horner -- the check used by radix2/mod.rs:320-360 test_fft_correctness */
void orc_fr_horner(const uint64_t *coeffs, size_t n, const uint64_t *x, uint64_t *out) {
    fr_t acc, xx;
    fr_zero(&acc);
    memcpy(xx.l, x, 32);
    for (size_t i = n; i-- > 0;) {
        fr_mul(&acc, &acc, &xx);
        fr_add(&acc, &acc, (const fr_t *)(coeffs + 4 * i));
    }
    memcpy(out, acc.l, 32);
}

/* ------------------------------------------------------------------ scalar-field element-wise exports */
#define ORC_BINOP(NAME, T, W, FN)                                                          \
    void NAME(const uint64_t *a, const uint64_t *b, uint64_t *out, size_t n) {             \
        for (size_t i = 0; i < n; i++) {                                                   \
            T r;                                                                           \
            FN(&r, (const T *)(a + W * i), (const T *)(b + W * i));                        \
            memcpy(out + W * i, &r, sizeof r);                                             \
        }                                                                                  \
    }
#define ORC_UNOP(NAME, T, W, FN)                                                           \
    void NAME(const uint64_t *a, uint64_t *out, size_t n) {                                \
        for (size_t i = 0; i < n; i++) {                                                   \
            T r;                                                                           \
            FN(&r, (const T *)(a + W * i));                                                \
            memcpy(out + W * i, &r, sizeof r);                                             \
        }                                                                                  \
    }
ORC_BINOP(orc_fr_mul, fr_t, 4, fr_mul)
ORC_BINOP(orc_fr_add, fr_t, 4, fr_add)
ORC_BINOP(orc_fr_sub, fr_t, 4, fr_sub)
ORC_UNOP(orc_fr_sqr, fr_t, 4, fr_sqr)
ORC_UNOP(orc_fr_neg, fr_t, 4, fr_neg)
ORC_UNOP(orc_fr_dbl, fr_t, 4, fr_dbl)
ORC_UNOP(orc_fr_inv, fr_t, 4, fr_inv)
ORC_BINOP(orc_fq_mul, fq_t, 6, fq_mul)
ORC_BINOP(orc_fq_add, fq_t, 6, fq_add)
ORC_BINOP(orc_fq_sub, fq_t, 6, fq_sub)
ORC_UNOP(orc_fq_sqr, fq_t, 6, fq_sqr)
ORC_UNOP(orc_fq_neg, fq_t, 6, fq_neg)
ORC_UNOP(orc_fq_dbl, fq_t, 6, fq_dbl)
ORC_UNOP(orc_fq_inv, fq_t, 6, fq_inv)
ORC_BINOP(orc_fq2_mul, fq2_t, 12, fq2_mul)
ORC_BINOP(orc_fq2_add, fq2_t, 12, fq2_add)
ORC_BINOP(orc_fq2_sub, fq2_t, 12, fq2_sub)
ORC_UNOP(orc_fq2_sqr, fq2_t, 12, fq2_sqr)
ORC_UNOP(orc_fq2_inv, fq2_t, 12, fq2_inv)

void orc_fr_into_repr(const uint64_t *a, uint64_t *out, size_t n) {
    for (size_t i = 0; i < n; i++) fr_into_repr(out + 4 * i, (const fr_t *)(a + 4 * i));
}
int orc_fr_from_repr(const uint64_t *a, uint64_t *out, size_t n) {
    for (size_t i = 0; i < n; i++) if (!fr_from_repr((fr_t *)(out + 4 * i), a + 4 * i)) return 1;
    return 0;
}
void orc_fq_into_repr(const uint64_t *a, uint64_t *out, size_t n) {
    for (size_t i = 0; i < n; i++) fq_into_repr(out + 6 * i, (const fq_t *)(a + 6 * i));
}
int orc_fq_from_repr(const uint64_t *a, uint64_t *out, size_t n) {
    for (size_t i = 0; i < n; i++) if (!fq_from_repr((fq_t *)(out + 6 * i), a + 6 * i)) return 1;
    return 0;
}

/* ------------------------------------------------------------------ group exports
 * Layouts (shared with include/czk.h): G1 affine = 12 u64 (x, y), G1 Jacobian = 18 u64 (x, y, z);
 * G2 affine = 24 u64 (x.c0, x.c1, y.c0, y.c1), G2 Jacobian = 36 u64.  Montgomery limbs. */
void orc_g1_msm(const uint64_t *bases, const uint8_t *inf, const uint64_t *scalars, size_t n, uint64_t *out_jac) {
    g1_jac_t r;
    g1_msm_pippenger(&r, (const g1_aff_t *)bases, inf, scalars, n);
    memcpy(out_jac, &r, sizeof r);
}
void orc_g2_msm(const uint64_t *bases, const uint8_t *inf, const uint64_t *scalars, size_t n, uint64_t *out_jac) {
    g2_jac_t r;
    g2_msm_pippenger(&r, (const g2_aff_t *)bases, inf, scalars, n);
    memcpy(out_jac, &r, sizeof r);
}
/* AffineCurve::multi_scalar_mul (algebra/ec/src/lib.rs:300-311): Montgomery scalars -> into_repr -> MSM */
void orc_g1_multi_scalar_mul(const uint64_t *bases, const uint8_t *inf, const uint64_t *scalars_mont, size_t n_bases,
                             size_t n_scalars, uint64_t *out_jac) {
    uint64_t *repr = (uint64_t *)malloc((n_scalars ? n_scalars : 1) * 32);
    orc_fr_into_repr(scalars_mont, repr, n_scalars);
    orc_g1_msm(bases, inf, repr, n_bases < n_scalars ? n_bases : n_scalars, out_jac);
    free(repr);
}
void orc_g2_multi_scalar_mul(const uint64_t *bases, const uint8_t *inf, const uint64_t *scalars_mont, size_t n_bases,
                             size_t n_scalars, uint64_t *out_jac) {
    uint64_t *repr = (uint64_t *)malloc((n_scalars ? n_scalars : 1) * 32);
    orc_fr_into_repr(scalars_mont, repr, n_scalars);
    orc_g2_msm(bases, inf, repr, n_bases < n_scalars ? n_bases : n_scalars, out_jac);
    free(repr);
}
int orc_g1_jac_to_affine(const uint64_t *jac, uint64_t *out_aff) {
    g1_aff_t a;
    int inf = g1_jac_to_affine(&a, (const g1_jac_t *)jac);
    memcpy(out_aff, &a, sizeof a);
    return inf;
}
int orc_g2_jac_to_affine(const uint64_t *jac, uint64_t *out_aff) {
    g2_aff_t a;
    int inf = g2_jac_to_affine(&a, (const g2_jac_t *)jac);
    memcpy(out_aff, &a, sizeof a);
    return inf;
}
void orc_g1_scalar_mul(const uint64_t *base_aff, int base_inf, const uint64_t *k_canonical, uint64_t *out_jac) {
    g1_jac_t r;
    g1_scalar_mul(&r, (const g1_aff_t *)base_aff, base_inf, k_canonical, 4);
    memcpy(out_jac, &r, sizeof r);
}
void orc_g2_scalar_mul(const uint64_t *base_aff, int base_inf, const uint64_t *k_canonical, uint64_t *out_jac) {
    g2_jac_t r;
    g2_scalar_mul(&r, (const g2_aff_t *)base_aff, base_inf, k_canonical, 4);
    memcpy(out_jac, &r, sizeof r);
}
void orc_g1_jac_add(const uint64_t *a, const uint64_t *b, uint64_t *out) {
    g1_jac_t r;
    memcpy(&r, a, sizeof r);
    g1_jac_add(&r, (const g1_jac_t *)b);
    memcpy(out, &r, sizeof r);
}
void orc_g2_jac_add(const uint64_t *a, const uint64_t *b, uint64_t *out) {
    g2_jac_t r;
    memcpy(&r, a, sizeof r);
    g2_jac_add(&r, (const g2_jac_t *)b);
    memcpy(out, &r, sizeof r);
}
void orc_g1_jac_add_mixed(const uint64_t *a, const uint64_t *b_aff, int b_inf, uint64_t *out) {
    g1_jac_t r;
    memcpy(&r, a, sizeof r);
    g1_jac_add_mixed(&r, (const g1_aff_t *)b_aff, b_inf);
    memcpy(out, &r, sizeof r);
}
void orc_g2_jac_add_mixed(const uint64_t *a, const uint64_t *b_aff, int b_inf, uint64_t *out) {
    g2_jac_t r;
    memcpy(&r, a, sizeof r);
    g2_jac_add_mixed(&r, (const g2_aff_t *)b_aff, b_inf);
    memcpy(out, &r, sizeof r);
}
void orc_g1_jac_double(const uint64_t *a, uint64_t *out) {
    g1_jac_t r;
    memcpy(&r, a, sizeof r);
    g1_jac_double(&r);
    memcpy(out, &r, sizeof r);
}
void orc_g2_jac_double(const uint64_t *a, uint64_t *out) {
    g2_jac_t r;
    memcpy(&r, a, sizeof r);
    g2_jac_double(&r);
    memcpy(out, &r, sizeof r);
}
/* y^2 == x^3 + b in Montgomery form (short_weierstrass_jacobian.rs is_on_curve) */
int orc_g1_on_curve(const uint64_t *aff) {
    const g1_aff_t *p = (const g1_aff_t *)aff;
    fq_t l, r, b;
    fq_sqr(&l, &p->y);
    fq_sqr(&r, &p->x);
    fq_mul(&r, &r, &p->x);
    fq_one(&b);                                   /* COEFF_B = 1 (curves/g1.rs:23) */
    fq_add(&r, &r, &b);
    return fq_eq(&l, &r);
}
int orc_g2_on_curve(const uint64_t *aff, const uint64_t *coeff_b) {
    const g2_aff_t *p = (const g2_aff_t *)aff;
    fq2_t l, r;
    fq2_sqr(&l, &p->y);
    fq2_sqr(&r, &p->x);
    fq2_mul(&r, &r, &p->x);
    fq2_add(&r, &r, (const fq2_t *)coeff_b);
    return fq2_eq(&l, &r);
}

/* ------------------------------------------------------------------ Groth16 per-party local compute
 * mpc-snarks/src/groth/r1cs_to_qap.rs:47-113 (witness_map) on ONE Fr lane.  `a`, `b`, `c` are D-element
 * buffers already holding the evaluated constraint rows (a[0..N) = <A_i, z>, a[N..N+n_inst) = z[0..n_inst),
 * zeros above; :67-83, :95-100).  The share-by-share product `F::batch_product_in_place(ab, b)` (:92) is a
 * communication step for shares; the caller supplies it through `beaver`:
 *   beaver == 0 : plain field product (single-prover flavour, proof.rs:75-110)
 *   beaver == 1 : the local half of Beaver multiplication with the dummy triple source
 *                 (mpc-algebra/src/share/field.rs:97-127, wire/field.rs:41-60) is NOT done here; callers that
 *                 model shares call orc_witness_map_pre / _post around their own open step.
 * Output h = ab (D elements). */
static void orc_witness_map_pre_lane(fr_t *a, fr_t *b, unsigned log_d) {
    orc_ntt_fr((uint64_t *)a, log_d, ORC_IFFT, (size_t)1 << log_d);
    orc_ntt_fr((uint64_t *)b, log_d, ORC_IFFT, (size_t)1 << log_d);
    orc_ntt_fr((uint64_t *)a, log_d, ORC_COSET_FFT, (size_t)1 << log_d);
    orc_ntt_fr((uint64_t *)b, log_d, ORC_COSET_FFT, (size_t)1 << log_d);
}
static void orc_witness_map_post_lane(fr_t *ab, fr_t *c, unsigned log_d) {
    size_t n = (size_t)1 << log_d;
    uint64_t consts[24];
    orc_domain_constants(log_d, consts);
    fr_t zinv;
    memcpy(zinv.l, consts + 20, 32);
    orc_ntt_fr((uint64_t *)c, log_d, ORC_IFFT, n);
    orc_ntt_fr((uint64_t *)c, log_d, ORC_COSET_FFT, n);
    for (size_t i = 0; i < n; i++) fr_sub(&ab[i], &ab[i], &c[i]);                 /* :105-107 */
    for (size_t i = 0; i < n; i++) fr_mul(&ab[i], &ab[i], &zinv);                 /* :109, domain/mod.rs:184-191 */
    orc_ntt_fr((uint64_t *)ab, log_d, ORC_COSET_IFFT, n);                          /* :110 */
}
void orc_witness_map_pre(uint64_t *a, uint64_t *b, unsigned log_d) { orc_witness_map_pre_lane((fr_t *)a, (fr_t *)b, log_d); }
void orc_witness_map_post(uint64_t *ab, uint64_t *c, unsigned log_d) { orc_witness_map_post_lane((fr_t *)ab, (fr_t *)c, log_d); }
/* single-prover witness map: a, b, c in; h written over a */
void orc_witness_map_plain(uint64_t *a, uint64_t *b, uint64_t *c, unsigned log_d) {
    size_t n = (size_t)1 << log_d;
    fr_t *fa = (fr_t *)a, *fb = (fr_t *)b;
    orc_witness_map_pre_lane(fa, fb, log_d);
    for (size_t i = 0; i < n; i++) fr_mul(&fa[i], &fa[i], &fb[i]);
    orc_witness_map_post_lane(fa, (fr_t *)c, log_d);
}

/* ------------------------------------------------------------------ all-host-cores CPU baseline (bench.py only)
 * Same values as the serial functions above; OpenMP tasks over independent pieces: the windows of an MSM (the
 * reference's `parallel` feature, variable_base.rs:33-37), the butterflies of one NTT stage, the independent a / b / c
 * chains and share lanes of the witness map, and the MSMs of a proof. */
/* `tw`: per-stage compacted roots -- the stage with butterfly span `gap` reads tw[gap - 1 + k] = root^(k * (n/2) / gap), k < gap
 * (what the serial code obtains by compacting its root vector after every stage, fft.rs:194-200): contiguous reads */
static fr_t *orc_stage_roots_par(size_t n, const fr_t *root) {
    fr_t *roots = orc_roots(n / 2, root);
    fr_t *tw = (fr_t *)malloc((n ? n : 1) * sizeof(fr_t));
    for (size_t gap = n / 2; gap > 0; gap /= 2) {
        const size_t stride = (n / 2) / gap;
#pragma omp taskloop grainsize(65536)
        for (size_t k = 0; k < gap; k++) tw[gap - 1 + k] = roots[k * stride];
    }
    free(roots);
    return tw;
}
static void orc_io_helper_par(fr_t *x, size_t n, const fr_t *tw) {
    for (size_t gap = n / 2; gap > 0; gap /= 2) {
        const fr_t *ts = tw + gap - 1;
#pragma omp taskloop grainsize(16384)
        for (size_t t = 0; t < n / 2; t++) {
            size_t base = (t / gap) * 2 * gap, k = t % gap;
            fr_t *lo = &x[base + k], *hi = &x[base + gap + k], neg;
            fr_sub(&neg, lo, hi);
            fr_add(lo, lo, hi);
            fr_mul(hi, &neg, &ts[k]);
        }
    }
}
static void orc_oi_helper_par(fr_t *x, size_t n, const fr_t *tw) {
    for (size_t gap = 1; gap < n; gap *= 2) {
        const fr_t *ts = tw + gap - 1;
#pragma omp taskloop grainsize(16384)
        for (size_t t = 0; t < n / 2; t++) {
            size_t base = (t / gap) * 2 * gap, k = t % gap;
            fr_t *lo = &x[base + k], *hi = &x[base + gap + k], neg;
            fr_mul(hi, hi, &ts[k]);
            fr_sub(&neg, lo, hi);
            fr_add(lo, lo, hi);
            *hi = neg;
        }
    }
}
/* pw[i] = c * g^i, in chunks (each chunk starts from one fr_pow) */
static void orc_scale_powers_par(fr_t *x, size_t n, const fr_t *g, const fr_t *c) {
    const size_t CH = 16384;
#pragma omp taskloop grainsize(1)
    for (size_t s = 0; s < (n + CH - 1) / CH; s++) {
        uint64_t e[1] = {(uint64_t)(s * CH)};
        fr_t pw;
        fr_pow(&pw, g, e, 1);
        fr_mul(&pw, &pw, c);
        size_t end = (s + 1) * CH < n ? (s + 1) * CH : n;
        for (size_t i = s * CH; i < end; i++) { fr_mul(&x[i], &x[i], &pw); fr_mul(&pw, &pw, g); }
    }
}
static void orc_derange_par(fr_t *x, size_t n, unsigned log_n) {
#pragma omp taskloop grainsize(65536)
    for (uint64_t idx = 1; idx < n - 1; idx++) {
        uint64_t r = 0, t = idx;
        for (unsigned b = 0; b < log_n; b++) { r = (r << 1) | (t & 1); t >>= 1; }
        if (idx < r) { fr_t tmp = x[idx]; x[idx] = x[r]; x[r] = tmp; }   /* each pair is swapped by its smaller index only */
    }
}
static void orc_ntt_fr_par(fr_t *x, unsigned log_d, int kind, const orc_domain_t *d, const fr_t *roots_fwd, const fr_t *roots_inv) {
    size_t n = (size_t)1 << log_d;
    fr_t one;
    fr_one(&one);
    if (kind == ORC_COSET_FFT) orc_scale_powers_par(x, n, &d->generator, &one);
    if (kind == ORC_FFT || kind == ORC_COSET_FFT) {
        orc_io_helper_par(x, n, roots_fwd);
        orc_derange_par(x, n, log_d);
    } else {
        orc_derange_par(x, n, log_d);
        orc_oi_helper_par(x, n, roots_inv);
        if (kind == ORC_IFFT) orc_scale_powers_par(x, n, &one, &d->size_inv);
        else orc_scale_powers_par(x, n, &d->generator_inv, &d->size_inv);
    }
}
/* One proof's local compute for `lanes` share lanes on all host cores: witness map (the product a * b stands in for the
 * Beaver local half: same multiplication count order) and the five MSMs per lane.  a, b, c: lanes x D evaluations (a ends
 * as h); wit: lanes x N, asg: lanes x (N+1) Montgomery scalars; out: lanes x (4 x 18 + 36) u64 (h, l, a, b_g1, b_g2). */
void orc_groth16_local_par(unsigned log_d, size_t N, size_t lanes, uint64_t *a, uint64_t *b, uint64_t *c, const uint64_t *wit,
                           const uint64_t *asg, const uint64_t *h_q, const uint64_t *l_q, const uint64_t *a_q, const uint64_t *b1_q,
                           const uint64_t *b2_q, const uint8_t *inf0, const uint8_t *inf_b, uint64_t *out, int threads) {
    const size_t D = (size_t)1 << log_d;
    orc_domain_t d;
    orc_domain_new(&d, D);
    uint64_t consts[24];
    orc_domain_constants(log_d, consts);
    fr_t zinv;
    memcpy(zinv.l, consts + 20, 32);
    fr_t *roots_fwd = orc_stage_roots_par(D, &d.group_gen), *roots_inv = orc_stage_roots_par(D, &d.group_gen_inv);   /* (outside a parallel region: built serially) */
    uint64_t *wit_r = (uint64_t *)malloc(lanes * N * 32), *asg_r = (uint64_t *)malloc(lanes * (N + 1) * 32);
    uint64_t *h_r = (uint64_t *)malloc(lanes * D * 32);
    if (threads > 0) omp_set_num_threads(threads);
    /* one bucket arena per thread, sized for the largest window (G2, c = window width of the longest MSM) */
    size_t cmax = (size_t)(orc_log2(D) * 69 / 100) + 2;
    size_t stride = ((((size_t)1 << cmax) * sizeof(g2_jac_t)) + 4095) & ~(size_t)4095;
    char *arena = (char *)malloc(stride * (size_t)omp_get_max_threads());
#pragma omp parallel
#pragma omp single
    {
#pragma omp taskgroup
        {
            for (size_t ln = 0; ln < lanes; ln++) {
                fr_t *la = (fr_t *)a + ln * D, *lb = (fr_t *)b + ln * D, *lc = (fr_t *)c + ln * D;
                const size_t O = 4 * 18 + 36;
                uint64_t *lo = out + ln * O;
#pragma omp task
                {   /* witness map -> h MSM (r1cs_to_qap.rs:85-110, prover.rs:104) */
#pragma omp taskgroup
                    {
#pragma omp task
                        { orc_ntt_fr_par(la, log_d, ORC_IFFT, &d, roots_fwd, roots_inv); orc_ntt_fr_par(la, log_d, ORC_COSET_FFT, &d, roots_fwd, roots_inv); }
#pragma omp task
                        { orc_ntt_fr_par(lb, log_d, ORC_IFFT, &d, roots_fwd, roots_inv); orc_ntt_fr_par(lb, log_d, ORC_COSET_FFT, &d, roots_fwd, roots_inv); }
#pragma omp task
                        { orc_ntt_fr_par(lc, log_d, ORC_IFFT, &d, roots_fwd, roots_inv); orc_ntt_fr_par(lc, log_d, ORC_COSET_FFT, &d, roots_fwd, roots_inv); }
                    }
#pragma omp taskloop grainsize(16384)
                    for (size_t i = 0; i < D; i++) {
                        fr_mul(&la[i], &la[i], &lb[i]);
                        fr_sub(&la[i], &la[i], &lc[i]);
                        fr_mul(&la[i], &la[i], &zinv);
                    }
                    orc_ntt_fr_par(la, log_d, ORC_COSET_IFFT, &d, roots_fwd, roots_inv);
#pragma omp taskloop grainsize(16384)
                    for (size_t i = 0; i < D; i++) fr_into_repr(h_r + 4 * (ln * D + i), &la[i]);
                    g1_jac_t r;
                    g1_msm_pippenger_par(&r, (const g1_aff_t *)h_q, inf0, h_r + 4 * ln * D, D - 1, arena, stride);
                    memcpy(lo, &r, sizeof r);
                }
#pragma omp task
                {   /* witness-only MSMs (prover.rs:108, 132-156) */
#pragma omp taskloop grainsize(16384)
                    for (size_t i = 0; i < N; i++) fr_into_repr(wit_r + 4 * (ln * N + i), (const fr_t *)(wit + 4 * (ln * N + i)));
#pragma omp taskloop grainsize(16384)
                    for (size_t i = 0; i < N + 1; i++) fr_into_repr(asg_r + 4 * (ln * (N + 1) + i), (const fr_t *)(asg + 4 * (ln * (N + 1) + i)));
#pragma omp taskgroup
                    {
#pragma omp task
                        { g2_jac_t r; g2_msm_pippenger_par(&r, (const g2_aff_t *)b2_q, inf_b, asg_r + 4 * ln * (N + 1), N + 1, arena, stride); memcpy(lo + 72, &r, sizeof r); }
#pragma omp task
                        { g1_jac_t r; g1_msm_pippenger_par(&r, (const g1_aff_t *)l_q, inf0, wit_r + 4 * ln * N, N, arena, stride); memcpy(lo + 18, &r, sizeof r); }
#pragma omp task
                        { g1_jac_t r; g1_msm_pippenger_par(&r, (const g1_aff_t *)a_q, inf0, asg_r + 4 * ln * (N + 1), N + 1, arena, stride); memcpy(lo + 36, &r, sizeof r); }
#pragma omp task
                        { g1_jac_t r; g1_msm_pippenger_par(&r, (const g1_aff_t *)b1_q, inf_b, asg_r + 4 * ln * (N + 1), N + 1, arena, stride); memcpy(lo + 54, &r, sizeof r); }
                    }
                }
            }
        }
    }
    free(roots_fwd); free(roots_inv); free(wit_r); free(asg_r); free(h_r); free(arena);
}
int orc_max_threads(void) { return omp_get_max_threads(); }
/* input generation for the CPU baseline: n distinct subgroup points (see ec_tmpl.h chain_points) */
void orc_g1_chain_points(const uint64_t *gen_aff, uint64_t *out, size_t n) { g1_chain_points((g1_aff_t *)out, n, (const g1_aff_t *)gen_aff); }
void orc_g2_chain_points(const uint64_t *gen_aff, uint64_t *out, size_t n) { g2_chain_points((g2_aff_t *)out, n, (const g2_aff_t *)gen_aff); }
/* FixedBaseMSM (algebra/ec/src/msm/fixed_base.rs): [k_i] * generator as affine points -- how the reference's generator derives a key's queries
 * (groth16/src/generator.rs:118-163); the checker for czk_fixed_base_points at full size and the base arrays of tests/golden/make_fullsize.py */
void orc_g1_fixed_base_msm(const uint64_t *gen_aff, const uint64_t *k, size_t n, uint64_t *out, uint8_t *out_inf, int threads) {
    if (threads > 0) omp_set_num_threads(threads);
    g1_fixed_base_msm((g1_aff_t *)out, out_inf, k, n, (const g1_aff_t *)gen_aff);
}
void orc_g2_fixed_base_msm(const uint64_t *gen_aff, const uint64_t *k, size_t n, uint64_t *out, uint8_t *out_inf, int threads) {
    if (threads > 0) omp_set_num_threads(threads);
    g2_fixed_base_msm((g2_aff_t *)out, out_inf, k, n, (const g2_aff_t *)gen_aff);
}

/* ------------------------------------------------------------------ MixedRadixEvaluationDomain<Fr> (size 2^a * 3^b, b <= 1)
 * algebra/poly/src/domain/mixed_radix.rs:232-262 -- mixed_radix_fft_permute */
static size_t orc_mixed_permute(unsigned two_adicity, unsigned q_adicity, size_t q, size_t n, size_t i) {
    size_t res = 0, shift = n;
    for (unsigned t = 0; t < two_adicity; t++) { shift /= 2; res += (i % 2) * shift; i /= 2; }
    for (unsigned t = 0; t < q_adicity; t++) { shift /= q; res += (i % q) * shift; i /= q; }
    return res;
}
static unsigned orc_k_adicity_fwd(size_t k, size_t n) {
    unsigned r = 0;
    while (n > 1) { if (n % k == 0) { r++; n /= k; } else return r; }
    return r;
}
int orc_fr_root_of_unity_mixed(size_t n, uint64_t *out);
/* mixed_radix.rs:286-404 -- serial_mixed_radix_fft (decimation in time: index permutation, the radix-q merge passes, then the
 * radix-2 merge passes), q = SMALL_SUBGROUP_BASE = 3 */
static void orc_serial_mixed_radix_fft(fr_t *a, size_t n, const fr_t *omega, unsigned two_adicity) {
    const size_t q = 3;
    unsigned q_adicity = orc_k_adicity_fwd(q, n);
    size_t m = 1;
    if (q_adicity > 0) {
        unsigned char *seen = (unsigned char *)calloc(n, 1);
        for (size_t k = 0; k < n; k++) {
            size_t i = k;
            fr_t a_i = a[i];
            while (!seen[i]) {
                size_t dest = orc_mixed_permute(two_adicity, q_adicity, q, n, i);
                fr_t a_dest = a[dest];
                a[dest] = a_i;
                seen[i] = 1;
                a_i = a_dest;
                i = dest;
            }
        }
        free(seen);
        fr_t omega_q, qth_roots[3], terms[2];
        uint64_t e[1] = {(uint64_t)(n / q)};
        fr_pow(&omega_q, omega, e, 1);
        fr_one(&qth_roots[0]);
        for (size_t i = 1; i < q; i++) fr_mul(&qth_roots[i], &qth_roots[i - 1], &omega_q);
        for (unsigned pass = 0; pass < q_adicity; pass++) {
            fr_t w_m;
            uint64_t em[1] = {(uint64_t)(n / (q * m))};
            fr_pow(&w_m, omega, em, 1);
            for (size_t k = 0; k < n; k += q * m) {
                fr_t w_j;
                fr_one(&w_j);
                for (size_t j = 0; j < m; j++) {
                    fr_t base_term = a[k + j], w_j_i = w_j;
                    for (size_t i = 1; i < q; i++) {
                        terms[i - 1] = a[k + j + i * m];
                        fr_mul(&terms[i - 1], &terms[i - 1], &w_j_i);
                        fr_mul(&w_j_i, &w_j_i, &w_j);
                    }
                    for (size_t i = 0; i < q; i++) {
                        a[k + j + i * m] = base_term;
                        for (size_t l = 1; l < q; l++) {
                            fr_t tmp;
                            fr_mul(&tmp, &terms[l - 1], &qth_roots[(i * l) % q]);
                            fr_add(&a[k + j + i * m], &a[k + j + i * m], &tmp);
                        }
                    }
                    fr_mul(&w_j, &w_j, &w_m);
                }
            }
            m *= q;
        }
    } else {
        for (size_t k = 0; k < n; k++) {
            size_t rk = 0, t = k;
            for (unsigned b = 0; b < two_adicity; b++) { rk = (rk << 1) | (t & 1); t >>= 1; }
            if (k < rk) { fr_t tmp = a[k]; a[k] = a[rk]; a[rk] = tmp; }
        }
    }
    for (unsigned pass = 0; pass < two_adicity; pass++) {
        fr_t w_m;
        uint64_t em[1] = {(uint64_t)(n / (2 * m))};
        fr_pow(&w_m, omega, em, 1);
        for (size_t k = 0; k < n; k += 2 * m) {
            fr_t w;
            fr_one(&w);
            for (size_t j = 0; j < m; j++) {
                fr_t t;
                fr_mul(&t, &a[(k + m) + j], &w);
                fr_sub(&a[(k + m) + j], &a[k + j], &t);
                fr_add(&a[k + j], &a[k + j], &t);
                fr_mul(&w, &w, &w_m);
            }
        }
        m *= 2;
    }
}
/* {fft, ifft, coset_fft, coset_ifft}_in_place of MixedRadixEvaluationDomain (mixed_radix.rs:130-157; coset_fft is the trait
 * default domain/mod.rs:139-142) on a buffer of `size` = 2^a or 3 * 2^a elements whose first in_len entries are the caller's
 * vector.  Returns 1 when no such domain exists. */
int orc_ntt_fr_mixed(uint64_t *data, size_t size, int kind, size_t in_len) {
    fr_t omega, omega_inv, sz, size_inv, gen, gen_inv, one;
    if (in_len > size || orc_fr_root_of_unity_mixed(size, (uint64_t *)omega.l)) return 1;
    unsigned two_adicity = orc_k_adicity_fwd(2, size);
    fr_inv(&omega_inv, &omega);
    fr_from_u64(&sz, size);
    fr_inv(&size_inv, &sz);
    memcpy(gen.l, fr_GENERATOR, sizeof gen.l);
    fr_inv(&gen_inv, &gen);
    fr_one(&one);
    fr_t *x = (fr_t *)data;
    if (kind == ORC_COSET_FFT) orc_distribute_powers(x, in_len, &gen, &one);
    for (size_t i = in_len; i < size; i++) fr_zero(&x[i]);
    if (kind == ORC_FFT || kind == ORC_COSET_FFT) {
        orc_serial_mixed_radix_fft(x, size, &omega, two_adicity);
    } else {
        orc_serial_mixed_radix_fft(x, size, &omega_inv, two_adicity);
        for (size_t i = 0; i < size; i++) fr_mul(&x[i], &x[i], &size_inv);
        if (kind == ORC_COSET_IFFT) orc_distribute_powers(x, size, &gen_inv, &one);
    }
    return 0;
}

/* ------------------------------------------------------------------ GSZ / Shamir open (test infrastructure)
 * algebra/ff/src/fields/utils.rs:3-14 -- k_adicity */
static unsigned orc_k_adicity(size_t k, size_t n) {
    unsigned r = 0;
    while (n > 1) {
        if (n % k == 0) { r++; n /= k; } else return r;
    }
    return r;
}
/* algebra/ff/src/fields/mod.rs:337-367 -- get_root_of_unity(n), LARGE_SUBGROUP branch in full (q = SMALL_SUBGROUP_BASE = 3,
 * SMALL_SUBGROUP_BASE_ADICITY = 1: fr.rs:19-20) */
int orc_fr_root_of_unity_mixed(size_t n, uint64_t *out) {
    const size_t q = 3;
    const unsigned small_subgroup_base_adicity = 1;
    unsigned q_adicity = orc_k_adicity(q, n);
    size_t q_part = 1;
    for (unsigned i = 0; i < q_adicity; i++) q_part *= q;
    unsigned two_adicity = orc_k_adicity(2, n);
    size_t two_part = (size_t)1 << two_adicity;
    if (n != two_part * q_part || two_adicity > FR_TWO_ADICITY || q_adicity > small_subgroup_base_adicity) return 1;   /* None */
    fr_t omega;
    memcpy(omega.l, fr_LARGE_ROOT, sizeof omega.l);
    const uint64_t qq[1] = {q};
    for (unsigned i = q_adicity; i < small_subgroup_base_adicity; i++) fr_pow(&omega, &omega, qq, 1);
    for (unsigned i = two_adicity; i < FR_TWO_ADICITY; i++) fr_sqr(&omega, &omega);
    memcpy(out, omega.l, 32);
    return 0;
}
/* mpc-algebra/src/share/gsz20/mod.rs:440-466 -- open_degree_vec over every element of a batch (batch_open :286-300):
 * shares: parties x n (party j's value of element i at shares[j][i]); ifft over MixedRadixEvaluationDomain::new(parties)
 * (mixed_radix.rs:141-151: the inverse DFT with group_gen_inv, then * size_inv -- evaluated here as the plain O(n^2) sum, the
 * same field values), `assert!(p.degree() <= d)` counted in *bad, result p.evaluate(0).  Returns 1 when no such domain exists. */
int orc_gsz_open(const uint64_t *shares, size_t parties, size_t n, const uint32_t *degrees, unsigned degree, uint64_t *out, uint64_t *bad) {
    fr_t w, winv, sz, size_inv;
    if (orc_fr_root_of_unity_mixed(parties, (uint64_t *)w.l)) return 1;
    fr_inv(&winv, &w);
    fr_from_u64(&sz, parties);
    fr_inv(&size_inv, &sz);
    fr_t *coef = (fr_t *)malloc(parties * sizeof(fr_t));
    *bad = 0;
    for (size_t i = 0; i < n; i++) {
        for (size_t k = 0; k < parties; k++) {
            fr_t acc, wk, pw;
            fr_zero(&acc);
            uint64_t e[1] = {k};
            fr_pow(&wk, &winv, e, 1);            /* (w^-1)^k */
            fr_one(&pw);
            for (size_t j = 0; j < parties; j++) {
                fr_t t;
                fr_mul(&t, (const fr_t *)(shares + 4 * (j * n + i)), &pw);
                fr_add(&acc, &acc, &t);
                fr_mul(&pw, &pw, &wk);
            }
            fr_mul(&coef[k], &acc, &size_inv);
        }
        /* DensePolynomial::from_coefficients_vec strips trailing zeros; degree() of the zero polynomial is 0 */
        size_t deg = 0;
        for (size_t k = parties; k-- > 0;) if (!fr_is_zero(&coef[k])) { deg = k; break; }
        unsigned d = degrees ? degrees[i] : degree;
        if (deg > d) (*bad)++;
        memcpy(out + 4 * i, coef[0].l, 32);      /* evaluate(0): Horner at zero leaves the constant coefficient */
    }
    free(coef);
    return 0;
}
/* share vector of a degree-`deg` polynomial with the given coefficients (deg + 1 Fr, Montgomery): out[j] = p(w^j), j < parties
 * -- how tests make t-shares (d.fft of the coefficient vector, as DomainCoeff does for MixedRadixEvaluationDomain) */
int orc_gsz_share(const uint64_t *coeffs, size_t n_coeffs, size_t parties, uint64_t *out) {
    fr_t w, x;
    if (orc_fr_root_of_unity_mixed(parties, (uint64_t *)w.l)) return 1;
    fr_one(&x);
    for (size_t j = 0; j < parties; j++) {
        orc_fr_horner(coeffs, n_coeffs, (const uint64_t *)x.l, out + 4 * j);
        fr_mul(&x, &x, &w);
    }
    return 0;
}

/* ---- callers either side of the NTT ("next" rows; test infrastructure like everything in this file) ------------- */
/* evaluate_constraint over every row of one R1CS matrix (mpc-snarks/src/groth/r1cs_to_qap.rs:12-42, 70-77, 95-100):
 * sum += (coeff == 1) ? val : val * coeff, in term order. */
void orc_r1cs_matvec(const uint64_t *row_ptr, const uint32_t *col, const uint64_t *coeff, size_t m, const uint64_t *z, uint64_t *out) {
    fr_t one;
    fr_one(&one);
    for (size_t i = 0; i < m; i++) {
        fr_t sum;
        memset(&sum, 0, sizeof sum);
        for (uint64_t t = row_ptr[i]; t < row_ptr[i + 1]; t++) {
            const fr_t *val = (const fr_t *)(z + 4 * (size_t)col[t]);
            const fr_t *k = (const fr_t *)(coeff + 4 * t);
            if (memcmp(k, &one, sizeof one) == 0) {
                fr_add(&sum, &sum, val);
            } else {
                fr_t prod;
                fr_mul(&prod, val, k);
                fr_add(&sum, &sum, &prod);
            }
        }
        memcpy(out + 4 * i, &sum, 32);
    }
}

/* DensePolynomial::divide_with_q_and_r (algebra/poly/src/polynomial/univariate/mod.rs:133-174) for the divisor
 * (X - z) = [-z, 1] that KZG10::compute_witness_polynomial builds (poly-commit/src/kzg10/mod.rs:205-209): schoolbook
 * long division from the top coefficient.  quotient: n-1 coefficients (zero-padded where the reference's vector is
 * shorter because of leading zeros), remainder: 1 coefficient. */
void orc_poly_div_linear(const uint64_t *coeffs, size_t n, const uint64_t *z, uint64_t *quotient, uint64_t *remainder) {
    fr_t negz, zero;
    memset(&zero, 0, sizeof zero);
    fr_neg(&negz, (const fr_t *)z);
    if (n > 1) memset(quotient, 0, (n - 1) * 32);
    memset(remainder, 0, 32);
    if (n == 0) return;
    fr_t *rem = malloc(n * sizeof(fr_t));
    memcpy(rem, coeffs, n * 32);
    size_t len = n;
    while (len && memcmp(&rem[len - 1], &zero, 32) == 0) len--;          /* from_coefficients_vec truncates */
    while (len >= 2) {                                                     /* degree >= divisor degree (1) */
        fr_t q = rem[len - 1];                                             /* * divisor_leading_inv (= 1) */
        size_t qdeg = len - 2;
        memcpy(quotient + 4 * qdeg, &q, 32);
        fr_t t;
        fr_mul(&t, &q, &negz);
        fr_sub(&rem[qdeg], &rem[qdeg], &t);                                /* i = 0: -= q * (-z) */
        fr_sub(&rem[qdeg + 1], &rem[qdeg + 1], &q);                        /* i = 1: -= q * 1    */
        while (len && memcmp(&rem[len - 1], &zero, 32) == 0) len--;
    }
    if (len == 1) memcpy(remainder, &rem[0], 32);
    free(rem);
}

/* the public running-product loop of partial_products (mpc-algebra/src/share/field.rs:169-172) */
void orc_fr_prefix_product(const uint64_t *x, size_t n, uint64_t *out) {
    fr_t last;
    for (size_t i = 0; i < n; i++) {
        fr_t v;
        memcpy(&v, x + 4 * i, 32);
        if (i) fr_mul(&v, &v, &last);
        last = v;
        memcpy(out + 4 * i, &v, 32);
    }
}

/* serial_batch_inversion_and_mul (algebra/ff/src/fields/mod.rs:642-677): v[i] <- coeff / v[i], zero elements skipped */
void orc_fr_batch_inverse(uint64_t *v, size_t n, const uint64_t *coeff) {
    fr_t *x = (fr_t *)v, zero, tmp, one;
    memset(&zero, 0, sizeof zero);
    fr_one(&one);
    fr_t *prod = malloc((n ? n : 1) * sizeof(fr_t));
    size_t np = 0;
    tmp = one;
    for (size_t i = 0; i < n; i++) {                                  /* first pass: [a, ab, abc, ...] over the non-zero elements */
        if (memcmp(&x[i], &zero, 32) == 0) continue;
        fr_mul(&tmp, &tmp, &x[i]);
        prod[np++] = tmp;
    }
    fr_inv(&tmp, &tmp);                                               /* guaranteed non-zero */
    fr_mul(&tmp, &tmp, (const fr_t *)coeff);
    size_t k = np;                                                    /* prod.rev().skip(1).chain(one) */
    for (size_t i = n; i-- > 0;) {
        if (memcmp(&x[i], &zero, 32) == 0) continue;
        fr_t s = k >= 2 ? prod[k - 2] : one, new_tmp;
        k--;
        fr_mul(&new_tmp, &tmp, &x[i]);
        fr_mul(&x[i], &tmp, &s);
        tmp = new_tmp;
    }
    free(prod);
}
