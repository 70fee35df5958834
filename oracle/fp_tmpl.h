/* TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's prime-field arithmetic.
 *
 * Included twice by czk_oracle.c: once with FP_N=4 / FP(x)=fr_##x (BLS12-377 Fr, Fp256) and once with
 * FP_N=6 / FP(x)=fq_##x (Fq, Fp384).  Every function states which reference lines it follows
 * (paths relative to /root/reference).  Elements are FP_N little-endian u64 limbs in Montgomery form,
 * always fully reduced (< p), exactly like `Fp256`/`Fp384` (algebra/ff/src/fields/macros.rs:103-108).
 *
 * Expects before inclusion: FP_N, FP(name), FP_T (element typedef name), and the constant arrays
 * FP(MODULUS), FP(R), FP(R2) plus FP(INV).
 */

typedef struct { uint64_t l[FP_N]; } FP_T;

/* biginteger/arithmetic.rs:5-37 -- adc / sbb / mac_with_carry on u128 */
static inline uint64_t FP(adc)(uint64_t a, uint64_t b, uint64_t *carry) {
    u128 t = (u128)a + b + *carry;
    *carry = (uint64_t)(t >> 64);
    return (uint64_t)t;
}
static inline uint64_t FP(sbb)(uint64_t a, uint64_t b, uint64_t *borrow) {
    u128 t = ((u128)1 << 64) + a - b - *borrow;
    *borrow = (t >> 64) == 0 ? 1 : 0;
    return (uint64_t)t;
}
static inline uint64_t FP(macc)(uint64_t a, uint64_t b, uint64_t c, uint64_t *carry) {
    u128 t = (u128)a + (u128)b * c + *carry;
    *carry = (uint64_t)(t >> 64);
    return (uint64_t)t;
}

static inline int FP(is_zero)(const FP_T *a) {
    uint64_t o = 0;
    for (int i = 0; i < FP_N; i++) o |= a->l[i];
    return o == 0;
}
static inline int FP(eq)(const FP_T *a, const FP_T *b) {
    uint64_t o = 0;
    for (int i = 0; i < FP_N; i++) o |= a->l[i] ^ b->l[i];
    return o == 0;
}
/* BigInteger Ord: compare from the most significant limb down */
static inline int FP(cmp_limbs)(const uint64_t *a, const uint64_t *b) {
    for (int i = FP_N - 1; i >= 0; i--) {
        if (a[i] < b[i]) return -1;
        if (a[i] > b[i]) return 1;
    }
    return 0;
}
static inline void FP(add_nocarry)(uint64_t *a, const uint64_t *b) {
    uint64_t c = 0;
    for (int i = 0; i < FP_N; i++) a[i] = FP(adc)(a[i], b[i], &c);
}
static inline void FP(sub_noborrow)(uint64_t *a, const uint64_t *b) {
    uint64_t br = 0;
    for (int i = 0; i < FP_N; i++) a[i] = FP(sbb)(a[i], b[i], &br);
}
/* macros.rs:237-246 -- is_valid / reduce: one conditional subtraction of p */
static inline void FP(reduce)(FP_T *a) {
    if (FP(cmp_limbs)(a->l, FP(MODULUS)) >= 0) FP(sub_noborrow)(a->l, FP(MODULUS));
}
static inline void FP(zero)(FP_T *a) { memset(a, 0, sizeof *a); }
static inline void FP(one)(FP_T *a) { memcpy(a->l, FP(R), sizeof a->l); } /* macros.rs:263-265 */
static inline int FP(is_one)(const FP_T *a) { return FP(cmp_limbs)(a->l, FP(R)) == 0; }

/* macros.rs:663-669 -- add_assign */
static inline void FP(add)(FP_T *r, const FP_T *a, const FP_T *b) {
    FP_T t = *a;
    FP(add_nocarry)(t.l, b->l);
    FP(reduce)(&t);
    *r = t;
}
/* macros.rs:672-680 -- sub_assign: add p first when b > a */
static inline void FP(sub)(FP_T *r, const FP_T *a, const FP_T *b) {
    FP_T t = *a;
    if (FP(cmp_limbs)(b->l, t.l) > 0) FP(add_nocarry)(t.l, FP(MODULUS));
    FP(sub_noborrow)(t.l, b->l);
    *r = t;
}
/* macros.rs:297-304 -- double_in_place: mul2 then reduce */
static inline void FP(dbl)(FP_T *r, const FP_T *a) {
    FP_T t;
    uint64_t top = 0;
    for (int i = 0; i < FP_N; i++) {
        t.l[i] = (a->l[i] << 1) | top;
        top = a->l[i] >> 63;
    }
    FP(reduce)(&t);
    *r = t;
}
/* macros.rs:605-617 -- neg: p - a, zero stays zero */
static inline void FP(neg)(FP_T *r, const FP_T *a) {
    if (FP(is_zero)(a)) { *r = *a; return; }
    FP_T t;
    memcpy(t.l, FP(MODULUS), sizeof t.l);
    FP(sub_noborrow)(t.l, a->l);
    *r = t;
}

/* fields/arithmetic.rs:7-56 -- mul_assign, the "no-carry" CIOS branch (both moduli qualify) */
static inline void FP(mul)(FP_T *out, const FP_T *a, const FP_T *b) {
    uint64_t r[FP_N];
    memset(r, 0, sizeof r);
    for (int i = 0; i < FP_N; i++) {
        uint64_t c1 = 0, c2 = 0;
        r[0] = FP(macc)(r[0], a->l[0], b->l[i], &c1);          /* fa::mac: carry-in is 0 */
        uint64_t k = r[0] * FP(INV);
        (void)FP(macc)(r[0], k, FP(MODULUS)[0], &c2);          /* fa::mac_discard */
        for (int j = 1; j < FP_N; j++) {
            r[j] = FP(macc)(r[j], a->l[j], b->l[i], &c1);
            r[j - 1] = FP(macc)(r[j], k, FP(MODULUS)[j], &c2);
        }
        r[FP_N - 1] = c1 + c2;
    }
    memcpy(out->l, r, sizeof r);
    FP(reduce)(out);
}

/* fields/arithmetic.rs:84-170 -- square_in_place: off-diagonal products, doubling, diagonal, then a
 * Montgomery reduction of the 2N-limb product. */
static inline void FP(sqr)(FP_T *out, const FP_T *a) {
    uint64_t r[2 * FP_N];
    memset(r, 0, sizeof r);
    uint64_t carry = 0;
    for (int i = 0; i < FP_N - 1; i++) {
        for (int j = i + 1; j < FP_N; j++) r[i + j] = FP(macc)(r[i + j], a->l[i], a->l[j], &carry);
        r[FP_N + i] = carry;
        carry = 0;
    }
    /* double the off-diagonal part: shift the whole 2N-limb value left by one bit */
    r[2 * FP_N - 1] = r[2 * FP_N - 2] >> 63;
    for (int i = 2 * FP_N - 2; i >= 2; i--) r[i] = (r[i] << 1) | (r[i - 1] >> 63);
    r[1] <<= 1;
    for (int i = 0; i < FP_N; i++) {
        r[2 * i] = FP(macc)(r[2 * i], a->l[i], a->l[i], &carry);
        r[2 * i + 1] = FP(adc)(r[2 * i + 1], 0, &carry);
    }
    uint64_t carry2 = 0;
    for (int i = 0; i < FP_N; i++) {
        uint64_t k = r[i] * FP(INV);
        uint64_t c = 0;
        (void)FP(macc)(r[i], k, FP(MODULUS)[0], &c);
        for (int j = 1; j < FP_N; j++) r[j + i] = FP(macc)(r[j + i], k, FP(MODULUS)[j], &c);
        r[FP_N + i] = FP(adc)(r[FP_N + i], carry2, &c);
        carry2 = c;
    }
    memcpy(out->l, r + FP_N, sizeof out->l);
    FP(reduce)(out);
}

/* fields/arithmetic.rs:59-81 -- into_repr: Montgomery reduction of (a, 0) => canonical limbs */
static inline void FP(into_repr)(uint64_t *out, const FP_T *a) {
    uint64_t r[FP_N];
    memcpy(r, a->l, sizeof r);
    for (int i = 0; i < FP_N; i++) {
        uint64_t k = r[i] * FP(INV);
        uint64_t c = 0;
        (void)FP(macc)(r[i], k, FP(MODULUS)[0], &c);
        for (int j = 1; j < FP_N; j++) r[(j + i) % FP_N] = FP(macc)(r[(j + i) % FP_N], k, FP(MODULUS)[j], &c);
        r[i % FP_N] = c;
    }
    memcpy(out, r, sizeof r);
}
/* macros.rs:443-454 -- from_repr: 0 stays 0; valid => times R2; otherwise None (returns 0) */
static inline int FP(from_repr)(FP_T *out, const uint64_t *repr) {
    FP_T t;
    memcpy(t.l, repr, sizeof t.l);
    if (FP(is_zero)(&t)) { *out = t; return 1; }
    if (FP(cmp_limbs)(t.l, FP(MODULUS)) >= 0) return 0;
    FP_T r2;
    memcpy(r2.l, FP(R2), sizeof r2.l);
    FP(mul)(out, &t, &r2);
    return 1;
}
static inline void FP(from_u64)(FP_T *out, uint64_t v) {
    uint64_t repr[FP_N];
    memset(repr, 0, sizeof repr);
    repr[0] = v;
    FP(from_repr)(out, repr);
}

static inline int FP(limbs_is_one)(const uint64_t *a) {
    if (a[0] != 1) return 0;
    for (int i = 1; i < FP_N; i++) if (a[i]) return 0;
    return 1;
}
static inline void FP(limbs_div2)(uint64_t *a) {
    uint64_t t = 0;
    for (int i = FP_N - 1; i >= 0; i--) {
        uint64_t t2 = a[i] << 63;
        a[i] = (a[i] >> 1) | t;
        t = t2;
    }
}
/* macros.rs:367-421 -- inverse: binary extended Euclid (Guajardo et al. Alg. 16), b starts at R2 */
static inline int FP(inv)(FP_T *out, const FP_T *a) {
    if (FP(is_zero)(a)) return 0;
    uint64_t u[FP_N], v[FP_N];
    memcpy(u, a->l, sizeof u);
    memcpy(v, FP(MODULUS), sizeof v);
    FP_T b, c;
    memcpy(b.l, FP(R2), sizeof b.l);
    FP(zero)(&c);
    while (!FP(limbs_is_one)(u) && !FP(limbs_is_one)(v)) {
        while ((u[0] & 1) == 0) {
            FP(limbs_div2)(u);
            if (b.l[0] & 1) FP(add_nocarry)(b.l, FP(MODULUS));
            FP(limbs_div2)(b.l);
        }
        while ((v[0] & 1) == 0) {
            FP(limbs_div2)(v);
            if (c.l[0] & 1) FP(add_nocarry)(c.l, FP(MODULUS));
            FP(limbs_div2)(c.l);
        }
        if (FP(cmp_limbs)(v, u) < 0) {
            FP(sub_noborrow)(u, v);
            FP(sub)(&b, &b, &c);
        } else {
            FP(sub_noborrow)(v, u);
            FP(sub)(&c, &c, &b);
        }
    }
    *out = FP(limbs_is_one)(u) ? b : c;
    return 1;
}

/* Field::pow (fields/mod.rs, square-and-multiply over the bits of a little-endian u64 exponent) */
static inline void FP(pow)(FP_T *out, const FP_T *base, const uint64_t *exp, int exp_limbs) {
    FP_T res;
    FP(one)(&res);
    int started = 0;
    for (int i = exp_limbs * 64 - 1; i >= 0; i--) {
        int bit = (exp[i / 64] >> (i % 64)) & 1;
        if (started) FP(sqr)(&res, &res);
        if (bit) {
            started = 1;
            FP(mul)(&res, &res, base);
        }
    }
    *out = res;
}
