/* TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's short-Weierstrass Jacobian group law and
 * of its Pippenger VariableBaseMSM.  Included twice by czk_oracle.c:
 *   G1: BF(x)=fq_##x,  BF_T=fq_t,  EC(x)=g1_##x
 *   G2: BF(x)=fq2_##x, BF_T=fq2_t, EC(x)=g2_##x
 * Paths relative to /root/reference.  a = 0 for both BLS12-377 groups (curves/g1.rs:38-40, g2.rs).
 */

typedef struct { BF_T x, y; } EC(aff_t);      /* the `infinity` flag travels as a separate byte */
typedef struct { BF_T x, y, z; } EC(jac_t);

/* short_weierstrass_jacobian.rs:444-457 -- zero() = (1, 1, 0); is_zero <=> z == 0 */
static void EC(jac_zero)(EC(jac_t) *p) { BF(one)(&p->x); BF(one)(&p->y); BF(zero)(&p->z); }
static int EC(jac_is_zero)(const EC(jac_t) *p) { return BF(is_zero)(&p->z); }

/* short_weierstrass_jacobian.rs:502-535 -- double_in_place, COEFF_A == 0 branch */
static void EC(jac_double)(EC(jac_t) *p) {
    if (EC(jac_is_zero)(p)) return;
    BF_T a, b, c, d, e, f, t;
    BF(sqr)(&a, &p->x);
    BF(sqr)(&b, &p->y);
    BF(sqr)(&c, &b);
    BF(add)(&t, &p->x, &b);
    BF(sqr)(&t, &t);
    BF(sub)(&t, &t, &a);
    BF(sub)(&t, &t, &c);
    BF(dbl)(&d, &t);
    BF(dbl)(&t, &a);          /* a.double_in_place() */
    BF(add)(&e, &a, &t);      /* e = a(old) + 2a */
    BF(sqr)(&f, &e);
    BF(mul)(&p->z, &p->z, &p->y);
    BF(dbl)(&p->z, &p->z);
    BF(sub)(&p->x, &f, &d);
    BF(sub)(&p->x, &p->x, &d);
    BF(sub)(&t, &d, &p->x);
    BF(mul)(&t, &t, &e);
    BF(dbl)(&c, &c); BF(dbl)(&c, &c); BF(dbl)(&c, &c);
    BF(sub)(&p->y, &t, &c);
}

/* short_weierstrass_jacobian.rs:570-638 -- add_assign_mixed (madd-2007-bl with explicit edge cases) */
static void EC(jac_add_mixed)(EC(jac_t) *p, const EC(aff_t) *q, int q_inf) {
    if (q_inf) return;
    if (EC(jac_is_zero)(p)) { p->x = q->x; p->y = q->y; BF(one)(&p->z); return; }
    BF_T z1z1, u2, s2;
    BF(sqr)(&z1z1, &p->z);
    BF(mul)(&u2, &q->x, &z1z1);
    BF(mul)(&s2, &q->y, &p->z);
    BF(mul)(&s2, &s2, &z1z1);
    if (BF(eq)(&p->x, &u2) && BF(eq)(&p->y, &s2)) { EC(jac_double)(p); return; }
    BF_T h, hh, i, j, r, v;
    BF(sub)(&h, &u2, &p->x);
    BF(sqr)(&hh, &h);
    BF(dbl)(&i, &hh); BF(dbl)(&i, &i);
    BF(mul)(&j, &h, &i);
    BF(sub)(&r, &s2, &p->y); BF(dbl)(&r, &r);
    BF(mul)(&v, &p->x, &i);
    BF(sqr)(&p->x, &r);
    BF(sub)(&p->x, &p->x, &j);
    BF(sub)(&p->x, &p->x, &v);
    BF(sub)(&p->x, &p->x, &v);
    BF(mul)(&j, &j, &p->y); BF(dbl)(&j, &j);
    BF(sub)(&p->y, &v, &p->x);
    BF(mul)(&p->y, &p->y, &r);
    BF(sub)(&p->y, &p->y, &j);
    BF(add)(&p->z, &p->z, &h);
    BF(sqr)(&p->z, &p->z);
    BF(sub)(&p->z, &p->z, &z1z1);
    BF(sub)(&p->z, &p->z, &hh);
}

/* short_weierstrass_jacobian.rs:666-728 -- add_assign (add-2007-bl with explicit edge cases) */
static void EC(jac_add)(EC(jac_t) *p, const EC(jac_t) *q) {
    if (EC(jac_is_zero)(p)) { *p = *q; return; }
    if (EC(jac_is_zero)(q)) return;
    BF_T z1z1, z2z2, u1, u2, s1, s2;
    BF(sqr)(&z1z1, &p->z);
    BF(sqr)(&z2z2, &q->z);
    BF(mul)(&u1, &p->x, &z2z2);
    BF(mul)(&u2, &q->x, &z1z1);
    BF(mul)(&s1, &p->y, &q->z); BF(mul)(&s1, &s1, &z2z2);
    BF(mul)(&s2, &q->y, &p->z); BF(mul)(&s2, &s2, &z1z1);
    if (BF(eq)(&u1, &u2) && BF(eq)(&s1, &s2)) { EC(jac_double)(p); return; }
    BF_T h, i, j, r, v, t;
    BF(sub)(&h, &u2, &u1);
    BF(dbl)(&i, &h); BF(sqr)(&i, &i);
    BF(mul)(&j, &h, &i);
    BF(sub)(&r, &s2, &s1); BF(dbl)(&r, &r);
    BF(mul)(&v, &u1, &i);
    BF(sqr)(&p->x, &r);
    BF(sub)(&p->x, &p->x, &j);
    BF(dbl)(&t, &v);
    BF(sub)(&p->x, &p->x, &t);
    BF(sub)(&t, &v, &p->x);
    BF(mul)(&t, &r, &t);
    BF(mul)(&s1, &s1, &j); BF(dbl)(&s1, &s1);
    BF(sub)(&p->y, &t, &s1);
    BF(add)(&t, &p->z, &q->z);
    BF(sqr)(&t, &t);
    BF(sub)(&t, &t, &z1z1);
    BF(sub)(&t, &t, &z2z2);
    BF(mul)(&p->z, &t, &h);
}

/* short_weierstrass_jacobian.rs:768-789 -- From<Projective> for Affine; returns the infinity flag */
static int EC(jac_to_affine)(EC(aff_t) *out, const EC(jac_t) *p) {
    if (EC(jac_is_zero)(p)) { BF(zero)(&out->x); BF(one)(&out->y); return 1; }   /* (0, 1, true) :149-151 */
    if (BF(is_one)(&p->z)) { out->x = p->x; out->y = p->y; return 0; }
    BF_T zi, zi2, zi3;
    BF(inv)(&zi, &p->z);
    BF(sqr)(&zi2, &zi);
    BF(mul)(&out->x, &p->x, &zi2);
    BF(mul)(&zi3, &zi2, &zi);
    BF(mul)(&out->y, &p->y, &zi3);
    return 0;
}

/* ProjectiveCurve::mul -- plain MSB-first double-and-add over a canonical little-endian scalar */
static void EC(scalar_mul)(EC(jac_t) *out, const EC(aff_t) *base, int base_inf, const uint64_t *k, int k_limbs) {
    EC(jac_t) acc;
    EC(jac_zero)(&acc);
    for (int i = k_limbs * 64 - 1; i >= 0; i--) {
        EC(jac_double)(&acc);
        if ((k[i / 64] >> (i % 64)) & 1) EC(jac_add_mixed)(&acc, base, base_inf);
    }
    *out = acc;
}

/* algebra/ec/src/msm/variable_base.rs:12-106 -- VariableBaseMSM::multi_scalar_mul, serial build.
 * scalars: n x 4 canonical limbs; bases: n affine points + infinity bytes.  Same window rule, same
 * unit-scalar shortcut, same bucket running sum, same high-to-low window fold. */
static void EC(msm_pippenger)(EC(jac_t) *out, const EC(aff_t) *bases, const uint8_t *inf,
                              const uint64_t *scalars, size_t size) {
    size_t c = size < 32 ? 3 : (size_t)(orc_log2(size) * 69 / 100) + 2;   /* :21-25, msm/mod.rs:10-13 */
    const size_t num_bits = 253;                                           /* FrParameters::MODULUS_BITS */
    const uint64_t fr_one[4] = {1, 0, 0, 0};                               /* one().into_repr() */
    size_t n_windows = (num_bits + c - 1) / c;
    size_t n_buckets = ((size_t)1 << c) - 1;
    EC(jac_t) *window_sums = (EC(jac_t) *)malloc(n_windows * sizeof(EC(jac_t)));
    EC(jac_t) *buckets = (EC(jac_t) *)malloc(n_buckets * sizeof(EC(jac_t)));
    for (size_t w = 0; w < n_windows; w++) {
        size_t w_start = w * c;
        EC(jac_t) res;
        EC(jac_zero)(&res);
        for (size_t b = 0; b < n_buckets; b++) EC(jac_zero)(&buckets[b]);
        for (size_t i = 0; i < size; i++) {
            const uint64_t *s = scalars + 4 * i;
            if ((s[0] | s[1] | s[2] | s[3]) == 0) continue;                /* filter(!s.is_zero()) :19 */
            if (memcmp(s, fr_one, sizeof fr_one) == 0) {
                if (w_start == 0) EC(jac_add_mixed)(&res, &bases[i], inf[i]);
                continue;
            }
            /* scalar.divn(w_start); scalar.as_ref()[0] % (1 << c)   (c <= 64 - always true here) */
            size_t limb = w_start / 64, off = w_start % 64;
            uint64_t lo = limb < 4 ? s[limb] >> off : 0;
            if (off && limb + 1 < 4) lo |= s[limb + 1] << (64 - off);
            uint64_t digit = lo & (((uint64_t)1 << c) - 1);
            if (digit) EC(jac_add_mixed)(&buckets[digit - 1], &bases[i], inf[i]);
        }
        EC(jac_t) running;
        EC(jac_zero)(&running);
        for (size_t b = n_buckets; b-- > 0;) {
            EC(jac_add)(&running, &buckets[b]);
            EC(jac_add)(&res, &running);
        }
        window_sums[w] = res;
    }
    EC(jac_t) total;
    EC(jac_zero)(&total);
    for (size_t w = n_windows - 1; w >= 1; w--) {
        EC(jac_add)(&total, &window_sums[w]);
        for (size_t k = 0; k < c; k++) EC(jac_double)(&total);
    }
    /* lowest + fold: `lowest + &...` is add_assign on a copy of lowest with the fold as `other` */
    EC(jac_t) lowest = window_sums[0];
    EC(jac_add)(&lowest, &total);
    *out = lowest;
    free(buckets);
    free(window_sums);
}

/* All-host-cores variant for the timed CPU baseline (bench.py cpu_baseline.all_cores): the same arithmetic with the
 * windows processed concurrently -- what the reference's `parallel` feature does with rayon
 * (variable_base.rs:33-37 `cfg_into_iter!(window_starts)`).  OpenMP tasks, so it composes with other tasks of an
 * enclosing parallel region; called outside one it runs serially. */
static void EC(msm_pippenger_par)(EC(jac_t) *out, const EC(aff_t) *bases, const uint8_t *inf,
                                  const uint64_t *scalars, size_t size, char *thread_scratch, size_t scratch_stride) {
    size_t c = size < 32 ? 3 : (size_t)(orc_log2(size) * 69 / 100) + 2;
    const size_t num_bits = 253;
    const uint64_t fr_one[4] = {1, 0, 0, 0};
    size_t n_windows = (num_bits + c - 1) / c;
    size_t n_buckets = ((size_t)1 << c) - 1;
    EC(jac_t) *window_sums = (EC(jac_t) *)malloc(n_windows * sizeof(EC(jac_t)));
#pragma omp taskloop grainsize(1) shared(window_sums)
    for (size_t w = 0; w < n_windows; w++) {
        /* bucket arrays come from a per-thread arena allocated once by the caller: 128 threads each malloc-ing and first-touching
         * ~10-20 MB per window serialise on the process's page-table lock */
        EC(jac_t) *buckets = thread_scratch && n_buckets * sizeof(EC(jac_t)) <= scratch_stride
                                 ? (EC(jac_t) *)(thread_scratch + scratch_stride * (size_t)omp_get_thread_num())
                                 : (EC(jac_t) *)malloc(n_buckets * sizeof(EC(jac_t)));
        const int own = !(thread_scratch && n_buckets * sizeof(EC(jac_t)) <= scratch_stride);
        size_t w_start = w * c;
        EC(jac_t) res;
        EC(jac_zero)(&res);
        for (size_t b = 0; b < n_buckets; b++) EC(jac_zero)(&buckets[b]);
        for (size_t i = 0; i < size; i++) {
            const uint64_t *s = scalars + 4 * i;
            if ((s[0] | s[1] | s[2] | s[3]) == 0) continue;
            if (memcmp(s, fr_one, sizeof fr_one) == 0) {
                if (w_start == 0) EC(jac_add_mixed)(&res, &bases[i], inf[i]);
                continue;
            }
            size_t limb = w_start / 64, off = w_start % 64;
            uint64_t lo = limb < 4 ? s[limb] >> off : 0;
            if (off && limb + 1 < 4) lo |= s[limb + 1] << (64 - off);
            uint64_t digit = lo & (((uint64_t)1 << c) - 1);
            if (digit) EC(jac_add_mixed)(&buckets[digit - 1], &bases[i], inf[i]);
        }
        EC(jac_t) running;
        EC(jac_zero)(&running);
        for (size_t b = n_buckets; b-- > 0;) {
            EC(jac_add)(&running, &buckets[b]);
            EC(jac_add)(&res, &running);
        }
        window_sums[w] = res;
        if (own) free(buckets);
    }
    EC(jac_t) total;
    EC(jac_zero)(&total);
    for (size_t w = n_windows - 1; w >= 1; w--) {
        EC(jac_add)(&total, &window_sums[w]);
        for (size_t k = 0; k < c; k++) EC(jac_double)(&total);
    }
    EC(jac_t) lowest = window_sums[0];
    EC(jac_add)(&lowest, &total);
    *out = lowest;
    free(window_sums);
}

/* P_0 = G, P_{i+1} = 2 P_i + G as affine points: n distinct subgroup points in O(n) group operations and ONE field
 * inversion per 1024 points (Montgomery's trick on the z coordinates) -- input generation for the CPU baseline only. */
static void EC(chain_points)(EC(aff_t) *out, size_t n, const EC(aff_t) *gen) {
    enum { CH = 1024 };
    EC(jac_t) *jac = (EC(jac_t) *)malloc(CH * sizeof(EC(jac_t)));
    BF_T *pre = (BF_T *)malloc(CH * sizeof(BF_T));
    EC(jac_t) cur;
    cur.x = gen->x; cur.y = gen->y; BF(one)(&cur.z);
    for (size_t start = 0; start < n; start += CH) {
        size_t m = n - start < CH ? n - start : CH;
        BF_T acc;
        BF(one)(&acc);
        for (size_t i = 0; i < m; i++) {
            jac[i] = cur;
            pre[i] = acc;
            BF(mul)(&acc, &acc, &cur.z);
            EC(jac_double)(&cur);
            EC(jac_add_mixed)(&cur, gen, 0);
        }
        BF_T inv;
        BF(inv)(&inv, &acc);
        for (size_t i = m; i-- > 0;) {
            BF_T zi, zi2, zi3;
            BF(mul)(&zi, &inv, &pre[i]);
            BF(mul)(&inv, &inv, &jac[i].z);
            BF(sqr)(&zi2, &zi);
            BF(mul)(&zi3, &zi2, &zi);
            BF(mul)(&out[start + i].x, &jac[i].x, &zi2);
            BF(mul)(&out[start + i].y, &jac[i].y, &zi3);
        }
    }
    free(jac);
    free(pre);
}

/* algebra/ec/src/msm/fixed_base.rs:11-96 -- FixedBaseMSM::{get_window_table, windowed_mul, multi_scalar_mul}, the way the
 * reference's generator builds a proving key's queries (groth16/src/generator.rs:118-163: scalar_size = Fr::size_in_bits() = 253,
 * window = get_mul_window_size(n) = ln_without_floats(n), 3 below 32 scalars).  out[i] = [k_i] g as AFFINE points + infinity bytes
 * (the generator's batch_normalization_into_affine, short_weierstrass_jacobian.rs:480-500).  k: n x 4 canonical limbs.  The scalars
 * are processed by all OpenMP threads (the reference's `cfg_iter!(v)`); each point's additions run in the reference's order. */
static void EC(fixed_base_msm)(EC(aff_t) *out, uint8_t *out_inf, const uint64_t *k, size_t n, const EC(aff_t) *g) {
    const size_t scalar_size = 253;
    size_t window = n < 32 ? 3 : (size_t)(orc_log2(n) * 69 / 100);      /* msm/mod.rs:10-13 ln_without_floats */
    if (window == 0) window = 1;
    const size_t in_window = (size_t)1 << window, outerc = (scalar_size + window - 1) / window;
    const size_t last_in_window = (size_t)1 << (scalar_size - (outerc - 1) * window);
    /* get_window_table (:20-58): multiples_of_g[outer][inner] = inner * 2^(window * outer) * g; entries beyond the last window's range stay zero */
    EC(aff_t) *table = (EC(aff_t) *)malloc(outerc * in_window * sizeof(EC(aff_t)));
    uint8_t *tinf = (uint8_t *)malloc(outerc * in_window);
    EC(jac_t) g_outer;
    g_outer.x = g->x; g_outer.y = g->y; BF(one)(&g_outer.z);
    for (size_t outer = 0; outer < outerc; outer++) {
        const size_t cur = outer == outerc - 1 ? last_in_window : in_window;
        EC(jac_t) g_inner;
        EC(jac_zero)(&g_inner);
        for (size_t inner = 0; inner < in_window; inner++) {
            EC(aff_t) *t = &table[outer * in_window + inner];
            if (inner < cur) {
                tinf[outer * in_window + inner] = (uint8_t)EC(jac_to_affine)(t, &g_inner);   /* batch_normalization_into_affine, element-wise here */
                EC(jac_add)(&g_inner, &g_outer);
            } else {
                tinf[outer * in_window + inner] = 1;
                BF(zero)(&t->x); BF(one)(&t->y);
            }
        }
        for (size_t i = 0; i < window; i++) EC(jac_double)(&g_outer);
    }
    /* multi_scalar_mul (:82-95) -> windowed_mul (:60-80) per scalar, then normalisation in chunks with one inversion each */
    enum { FB_CH = 1024 };
#pragma omp parallel
    {
        EC(jac_t) *jac = (EC(jac_t) *)malloc(FB_CH * sizeof(EC(jac_t)));
        BF_T *pre = (BF_T *)malloc(FB_CH * sizeof(BF_T));
#pragma omp for schedule(dynamic, 1)
        for (size_t start = 0; start < n; start += FB_CH) {
            const size_t m = n - start < FB_CH ? n - start : FB_CH;
            BF_T acc;
            BF(one)(&acc);
            for (size_t i = 0; i < m; i++) {
                const uint64_t *s = k + 4 * (start + i);
                EC(jac_t) res;
                EC(jac_zero)(&res);                                     /* multiples_of_g[0][0].into_projective() = zero */
                for (size_t outer = 0; outer < outerc; outer++) {
                    size_t inner = 0;
                    for (size_t b = 0; b < window; b++) {
                        const size_t bit = outer * window + b;
                        if (bit < 253 && ((s[bit / 64] >> (bit % 64)) & 1)) inner |= (size_t)1 << b;   /* bit < modulus_size (:72) */
                    }
                    EC(jac_add_mixed)(&res, &table[outer * in_window + inner], tinf[outer * in_window + inner]);
                }
                jac[i] = res;
                pre[i] = acc;
                if (!EC(jac_is_zero)(&res)) BF(mul)(&acc, &acc, &res.z);
            }
            BF_T inv;
            BF(inv)(&inv, &acc);
            for (size_t i = m; i-- > 0;) {
                if (EC(jac_is_zero)(&jac[i])) {
                    out_inf[start + i] = 1;
                    BF(zero)(&out[start + i].x); BF(one)(&out[start + i].y);
                    continue;
                }
                BF_T zi, zi2, zi3;
                BF(mul)(&zi, &inv, &pre[i]);
                BF(mul)(&inv, &inv, &jac[i].z);
                BF(sqr)(&zi2, &zi);
                BF(mul)(&zi3, &zi2, &zi);
                BF(mul)(&out[start + i].x, &jac[i].x, &zi2);
                BF(mul)(&out[start + i].y, &jac[i].y, &zi3);
                out_inf[start + i] = 0;
            }
        }
        free(jac);
        free(pre);
    }
    free(table);
    free(tinf);
}
