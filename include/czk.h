/* czk.h -- C ABI of libczk_hip.so: the MI355X (gfx950) implementation of the per-party local compute
 * of collaborative-zksnark's MPC provers (radix-2 NTT over BLS12-377 Fr on share lanes; variable-base MSM
 * over BLS12-377 G1/G2).  Plain pointers and sizes only -- no torch / HIP types in any signature.
 *
 * The reference (alex-ozdemir/collaborative-zksnark) has no FFI; its seams are Rust trait impls.  Each
 * entry point below names the reference interface it replaces (paths relative to the reference root).
 * INTEGRATION.md shows the Rust `extern "C"` shim a maintainer would add at each seam.
 *
 * Data formats (identical to the reference's in-memory values):
 *   Fr  : 4 x u64 little-endian limbs, Montgomery form R = 2^256   (algebra/ff/src/fields/macros.rs:103-108,
 *         curves/bls12_377/src/fields/fr.rs:48-53); "canonical" = `into_repr()` BigInteger256.
 *   Fq  : 6 x u64, Montgomery form R = 2^384                         (curves/bls12_377/src/fields/fq.rs:43-50)
 *   Fq2 : c0, c1 (12 x u64)                                           (fields/models/quadratic_extension.rs:136-142)
 *   G1 affine  : x, y          = 12 u64  + a separate u8 infinity flag per point
 *   G1 Jacobian: x, y, z       = 18 u64  (infinity <=> z == 0)        (short_weierstrass_jacobian.rs:338-344)
 *   G2 affine  : x.c0 x.c1 y.c0 y.c1 = 24 u64 + u8 flag;  G2 Jacobian = 36 u64
 * Rust's struct layout is unspecified (no #[repr(C)]), so the shim repacks into these arrays.
 *
 * Memory: every buffer argument is either host memory (CZK_MEM_HOST: the library stages it through
 * HBM) or device memory on the context's GPU (CZK_MEM_DEVICE: used in place, no copies).
 * Threading: one czk_ctx = one GPU + one HIP stream = one MPC party; calls on a ctx are serialized by
 * the caller (the reference prover is single-threaded; mpc-net/src/multi.rs:15-23).
 * Errors: every call returns a czk_status; the reference's `None`/`assert!`/`unwrap()` sites map to
 * CZK_ERR_SIZE / CZK_ERR_ARG and the shim `expect()`s them, preserving panic behaviour.
 */
#ifndef CZK_H
#define CZK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct czk_ctx czk_ctx;
typedef struct czk_bases czk_bases;

typedef enum {
    CZK_OK = 0,
    CZK_ERR_SIZE = 1,  /* domain too large (log2 D > TWO_ADICITY = 47: radix2/mod.rs:61-63) or len > D (:100) */
    CZK_ERR_HIP = 2,   /* HIP runtime failure (message via czk_last_error) */
    CZK_ERR_ARG = 3,   /* null / inconsistent argument */
    CZK_ERR_NOMEM = 4,
    CZK_ERR_NET = 5,   /* czk_net_*: transport failure or a peer that did not arrive within the communicator's timeout (message via czk_net_last_error) */
    CZK_ERR_CHECK = 6  /* czk_net_atomic_broadcast: a party's data does not match its commitment (the reference's assert_eq!, channel.rs:63-66) */
} czk_status;

typedef enum {
    CZK_MEM_HOST = 0,
    CZK_MEM_DEVICE = 1,
    /* czk_msm_async only, OR-ed with CZK_MEM_DEVICE: the caller will not modify the scalar buffer before the next
     * czk_ctx_sync(), so the context's stream need not wait for the MSM's digit extraction (lets independent work
     * enqueued afterwards on that stream -- e.g. the witness-map NTTs -- overlap with the MSM). */
    CZK_MEM_STABLE = 16,
    /* czk_bases_register only, OR-ed with CZK_MEM_HOST / CZK_MEM_DEVICE: keep the points only, no precomputed window
     * multiples.  Registration is then a copy (instead of ~240 point doublings per point: 65 ms per 2^20 G1 points), and
     * each MSM runs one bucket set per window plus the Horner combination of variable_base.rs:92-105 -- about 1.4x the
     * arithmetic of the table form.  For callers that use a set of bases once or a few times; czk_msm_g1 / czk_msm_g2 use it. */
    CZK_MEM_NO_TABLES = 32,
    /* czk_bases_register only, OR-ed in: the bases are arbitrary points of the curve, not necessarily elements of the prime-order
     * subgroup.  G1 handles then keep short-Weierstrass (XYZZ) bucket arithmetic with the reference's complete case analysis
     * (short_weierstrass_jacobian.rs:570-597).  Without the flag G1 bases are taken to lie in G1 -- as every proving-key and SRS element
     * does (the reference deserialises them with a subgroup check) -- and the bucket kernels use the curve's twisted Edwards form, whose
     * unified 7-multiplication addition is exception-free exactly on that subgroup.  Results are the same group elements either way.
     * A base of even order registered WITHOUT this flag (and without the check below) can meet an exceptional pair of the unified law
     * and yield a wrong group element with status CZK_OK: a caller that cannot vouch for its bases passes one of the two flags. */
    CZK_MEM_ANY_POINTS = 64,
    /* czk_bases_register only, OR-ed in: verify the assumption instead of trusting it.  Registration runs the reference's
     * is_in_correct_subgroup_assuming_on_curve ([r] P == infinity, short_weierstrass_jacobian.rs:131; plus y^2 = x^3 + b) on every base
     * -- what the reference does when it deserialises a key (:868, :881) -- and keeps the handle on the complete XYZZ kernels when any
     * base fails, exactly as CZK_MEM_ANY_POINTS would.  czk_bases_check_subgroup then reports the count without re-running.  Costs about
     * as much as building the window tables (252 doublings + 87 additions per point; 2^20 G1 points: ~0.1 s). */
    CZK_MEM_CHECK_SUBGROUP = 128,
    /* czk_fr_vec_scale only, OR-ed with CZK_MEM_DEVICE: the vectors are device memory, the scalar `k` is HOST memory, read when the call is
     * made and handed to the kernel with the launch -- no device copy of a 32-byte value, no lifetime to observe. */
    CZK_MEM_SCALAR_HOST = 256,
    /* czk_msm_async only, OR-ed with CZK_MEM_DEVICE | CZK_MEM_STABLE: the scalars are those of the most recent czk_msm_async call on this context that
     * did NOT carry this flag (the leader of a group of calls over one vector) -- same pointer, count, lanes and form, that call CZK_MEM_STABLE too,
     * the buffer unchanged -- as when one assignment vector meets several queries (Groth16's a, b_g1, b_g2: groth16/src/prover.rs:130-166 passes
     * `assignment` three times).  The library may then read a digit sort made by the leader (or by a later member of the group) instead of sorting
     * again; it does while that sort is still in one of the pipeline's workspace slots and both keys run the same table layout (size, window width)
     * and have the same points at infinity, otherwise the flag changes nothing (nor does it unless the option "msm_sort_reuse" is set).  Results are the same either way.  Nothing is carried from one
     * group to the next or across czk_ctx_sync / czk_ctx_wait_mark: the next proof's first call over the same buffer sorts again. */
    CZK_MEM_SAME_SCALARS = 512
} czk_mem;

/* EvaluationDomain::{fft, ifft, coset_fft, coset_ifft}_in_place (algebra/poly/src/domain/mod.rs:79,90,139,155) */
typedef enum { CZK_FFT = 0, CZK_IFFT = 1, CZK_COSET_FFT = 2, CZK_COSET_IFFT = 3 } czk_ntt_kind;

/* scalar encodings accepted by the MSM */
typedef enum {
    CZK_SCALAR_CANONICAL = 0, /* BigInteger256, as VariableBaseMSM::multi_scalar_mul takes (msm/variable_base.rs:12-15) */
    CZK_SCALAR_MONTGOMERY = 1 /* Fr, as AffineCurve::multi_scalar_mul takes; into_repr runs on the GPU (ec/src/lib.rs:300-311) */
} czk_scalar_form;

typedef enum { CZK_G1 = 1, CZK_G2 = 2 } czk_group;

/* ---- context ------------------------------------------------------------------------------------ */
/* `hip_stream`: a hipStream_t to enqueue on (e.g. torch's current stream), or NULL for a private one. */
int czk_ctx_create(czk_ctx** out, int device, void* hip_stream);
void czk_ctx_destroy(czk_ctx* ctx);
int czk_ctx_sync(czk_ctx* ctx);
/* Waiting for PART of a context's work.  The reference's polynomial provers stop at every transcript point for what the transcript absorbs there
 * (the commitments and evaluations sent so far: mpc-plonk/src/lib.rs:430-448, marlin/src/lib.rs:176-318) -- not for everything the prover has
 * started: work that depends on no pending challenge (the next round's transforms of public data, commitments of polynomials that are already
 * known) can be enqueued before the wait and runs through it.  czk_ctx_mark names the work enqueued on the context so far -- kernels on its stream,
 * czk_msm_async calls, czk_lanes_download_deferred copies; czk_ctx_wait_mark blocks until THAT work is done and delivers its host results (what
 * czk_ctx_sync does for all work), while calls made after the mark keep running.  A mark covers the marks taken before it; waiting for a retired mark
 * returns at once; czk_ctx_sync retires every mark. */
int czk_ctx_mark(czk_ctx* ctx, uint64_t* out_mark);
int czk_ctx_wait_mark(czk_ctx* ctx, uint64_t mark);
/* The hipStream_t the context enqueues on (the one given to czk_ctx_create, or its private stream): lets a caller order its own
 * streams against the context's with events (hipStreamWaitEvent) instead of czk_ctx_sync -- e.g. an RCCL exchange between two opens. */
void* czk_ctx_stream(const czk_ctx* ctx);
/* What the library otherwise allocates and builds on FIRST use, done now: the twiddle / coset tables of the radix-2 domain 2^ntt_log_d and the
 * pass scratch for `ntt_lanes` lanes (ntt_lanes = 0: skip), and -- for MSMs of n_scalars scalars x msm_lanes lanes over `bases` (NULL: skip) --
 * the MSM pipeline's streams, the table set such calls use and every workspace of its ring.  A prover that cares about the latency of its first
 * proof after loading a key calls this once per (domain, key); results are unaffected.  Blocking. */
int czk_ctx_reserve(czk_ctx* ctx, unsigned ntt_log_d, size_t ntt_lanes, const czk_bases* bases, size_t n_scalars, size_t msm_lanes);
const char* czk_last_error(const czk_ctx* ctx);
const char* czk_version(void);
/* Tuning options of one context.  The library reads NO environment variables: everything a caller may select is named here.
 *   "msm_slots" 1..4            depth of the MSM pipeline's workspace ring (default 4)                          [before the first MSM]
 *   "msm_stream_priority" 0..2  0 = equal priorities (default), 1 = sort / reduce streams above accumulate, 2 = the reverse [before the first MSM]
 *   "msm_lane_interleave" 0..64 lanes per interleave group of the bucket accumulation kernels: neighbouring threads take the same bucket rank of G
 *                               neighbouring share lanes (1 = one lane per workgroup row, the layout of rounds 1 - 5; 0 = the library's default: 4 for keys
 *                               with window tables, 1 on the table-free path)
 *   "msm_sort_reuse" 0/1        honour CZK_MEM_SAME_SCALARS between keys registered AFTERWARDS (default 0 = every call sorts: sharing b_g2's sort with b_g1 measured + 0.5 % per proof when the two
 *                               calls are neighbours, - 0.6 % two calls apart -- the sorts run beside the accumulate kernels anyway, and entries sorted a moment
 *                               ago are still in the last-level cache when their own accumulate kernel reads them)
 *   "msm_sort_onepass" 0/1      single-pass digit sort for every call (default: only beyond 2048 partitions)
 *   "msm_fixed_c" 0/1           keys registered AFTERWARDS keep their own window width for short calls (no secondary table sets)
 *   "msm_window_g1" / "msm_window_g2" 0, 8..22   primary window width of keys registered AFTERWARDS (0 = the cost model, default)
 *   "ntt_gen1" 0/1              first-generation NTT passes (the small-domain kernels) for every size
 *   "ntt_fuse_pairs" 0/1        czk_witness_map_pre/post: the last pass of an inverse transform and the first pass of the coset transform that follows it
 *                               as one kernel where their tiles line up (2^21, 2^18, 2^15, 2^14, 2^12); same values; default 0 (+ 0.5 % per proof measured)
 *   "net_create_timeout_ms" 0..3600000   how long czk_net_create on this context waits for its peers at the rendezvous (0 = the default, 120 s;
 *                               afterwards the communicator's own "timeout_ms" option applies)
 * Any other name is CZK_ERR_ARG.  The call drains the context's enqueued work first, so an option never changes under a running proof.
 * (libczk_hip_lab.so, the -DCZK_LAB build of the same sources, additionally knows the switches of the measured-and-rejected kernel
 * variants it alone contains -- EXPERIMENTS.md; czk_build_is_lab() tells the two apart.) */
int czk_ctx_set_option(czk_ctx* ctx, const char* name, long value);
int czk_build_is_lab(void);

/* ---- device-resident share lanes ---------------------------------------------------------------------- */
/* A caller without a HIP allocator of its own (the Rust shim, a C++ host) keeps share vectors on the GPU between calls through
 * these handles: R1CStoQAP::witness_map holds `a`, `b`, `c` across seven transforms and hands `h` straight to the MSM
 * (mpc-snarks/src/groth/r1cs_to_qap.rs:85-110, mpc-snarks/src/groth/prover.rs:104) -- with a handle none of that crosses PCIe.
 * A czk_lanes is `lanes` x `len` Fr (lane-major, 4 u64 per element: the layout every `lanes x D x 4 u64` argument of this
 * header has), allocated zero-filled (`vec![T::zero(); domain_size]`, r1cs_to_qap.rs:66-67) on the context's GPU.
 * czk_lanes_data(l, lane, elem) is the address of one element: pass it wherever a buffer argument is taken with
 * CZK_MEM_DEVICE (czk_ntt_fr, czk_fr_vec_op, czk_witness_map_pre/post, czk_r1cs_matvec, czk_msm(_async), ...).  NULL when
 * (lane, elem) is outside the allocation.  The handle must be freed before its context is destroyed.
 * czk_lanes_upload: n Fr from pageable HOST memory (a Rust Vec) into lane `lane` at element `elem`, staged through the context's
 * pinned buffers; returns once `host` has been read (the caller may drop the Vec), the last DMA completes in stream order
 * before any later call on the context (a run may continue into the following lanes: the array is contiguous).  czk_lanes_download: the reverse, blocking (the values are in `host` on return).
 * czk_lanes_copy: device-to-device, in stream order (`let mut ab = a.clone()`).  CZK_ERR_ARG when a range leaves the lanes. */
typedef struct czk_lanes czk_lanes;
int czk_lanes_alloc(czk_ctx* ctx, size_t lanes, size_t len, czk_lanes** out);
void czk_lanes_free(czk_lanes* l);
size_t czk_lanes_count(const czk_lanes* l);
size_t czk_lanes_len(const czk_lanes* l);
uint64_t* czk_lanes_data(const czk_lanes* l, size_t lane, size_t elem);
int czk_lanes_upload(czk_ctx* ctx, czk_lanes* dst, size_t lane, size_t elem, const uint64_t* host, size_t n);
int czk_lanes_download(czk_ctx* ctx, const czk_lanes* src, size_t lane, size_t elem, uint64_t* host, size_t n);
/* czk_lanes_download without the wait: the copy is enqueued in stream order (through the library's pinned result staging, at most 4 MiB pending) and
 * `host` is filled when a mark taken after this call is waited for (czk_ctx_wait_mark) or at the next czk_ctx_sync -- the delivery rule of
 * czk_msm_async's results.  `host` must stay valid until then. */
int czk_lanes_download_deferred(czk_ctx* ctx, const czk_lanes* src, size_t lane, size_t elem, uint64_t* host, size_t n);
int czk_lanes_copy(czk_ctx* ctx, czk_lanes* dst, size_t dst_lane, size_t dst_elem, const czk_lanes* src, size_t src_lane, size_t src_elem,
                   size_t n);
int czk_lanes_zero(czk_ctx* ctx, czk_lanes* dst, size_t lane, size_t elem, size_t n);

/* Strided copy / zero fill of Fr elements in DEVICE memory, in stream order: for i0 < n[0], i1 < n[1], i2 < n[2]
 *     dst[i0 * dst_stride[0] + i1 * dst_stride[1] + i2 * dst_stride[2]] = src[i0 * src_stride[0] + i1 * src_stride[1] + i2 * src_stride[2]]
 * (strides in Fr elements; src NULL: the destination elements become zero).  One kernel for every re-layout a prover's polynomial code does
 * between transforms -- `resize`, `coeffs[k..].to_vec()`, concatenation of coefficient ranges, stacking share lanes, and the interleave /
 * de-interleave by a stride of mpc-plonk's `shift` products (mpc-plonk/src/lib.rs:150-200) and Marlin's matrix polynomials
 * (marlin/src/ahp/prover.rs:500-640) -- so a host needs no tensor library for them.  Source and destination ranges must not overlap. */
int czk_fr_copy_3d(czk_ctx* ctx, uint64_t* dst, const size_t* dst_stride, const uint64_t* src, const size_t* src_stride, const size_t* n);

/* ---- NTT ---------------------------------------------------------------------------------------- */
/* Replaces Radix2EvaluationDomain<Fr>::{fft,ifft,coset_ifft}_in_place (algebra/poly/src/domain/radix2/mod.rs:99-117)
 * and the trait-default coset_fft_in_place (domain/mod.rs:139-142) for T = Fr lanes (an MpcField<Fr, SpdzFieldShare>
 * vector is two lanes: sh and mac; mpc-algebra/src/share/spdz.rs:186-202).
 * data: lanes x D x 4 u64, D = 2^log_d, lane-major, transformed in place, natural order in and out.
 * in_len <= D: entries [in_len, D) of every lane are taken as zero (`resize(size, T::zero())`) whatever they hold.
 * Root of unity = get_root_of_unity(D) = LARGE_SUBGROUP_ROOT_OF_UNITY^3 squared down (ff/src/fields/mod.rs:360-367);
 * coset shift = Fr::multiplicative_generator() = 22.
 * On an error return `data` is undefined (host memory: lanes are written back as they finish, so some may already hold their
 * transform); the reference panics at the corresponding points, so no caller continues with the vector. */
int czk_ntt_fr(czk_ctx* ctx, uint64_t* data, unsigned log_d, size_t lanes, int kind, size_t in_len, int mem);
/* The non-in-place forms EvaluationDomain::{fft, ifft, coset_fft, coset_ifft}(&self, coeffs: &[T]) -> Vec<T> (algebra/poly/src/domain/mod.rs:
 * 72-76, 83-87, 130-134, 146-150: `coeffs.to_vec()` then the in-place transform): reads in_len elements per lane from src (lane k at
 * src + 4 * k * src_stride u64) and writes the 2^log_d results per lane to dst (lane stride 2^log_d); the copy the reference makes is the
 * transform's first load.  src and dst must not overlap unless they are equal with src_stride = 2^log_d.  DEVICE memory only. */
int czk_ntt_fr_to(czk_ctx* ctx, const uint64_t* src, size_t src_stride, uint64_t* dst, unsigned log_d, size_t lanes, int kind,
                  size_t in_len, int mem);

/* Domain constants as the reference's Radix2EvaluationDomain::new computes them (radix2/mod.rs:51-82), Montgomery
 * limbs: out[0..4) size_inv, [4..8) group_gen, [8..12) group_gen_inv, [12..16) generator (22), [16..20) generator_inv,
 * [20..24) (generator^D - 1)^-1 used by divide_by_vanishing_poly_on_coset_in_place (domain/mod.rs:184-191). */
int czk_domain_constants(czk_ctx* ctx, unsigned log_d, uint64_t* out24);

/* MixedRadixEvaluationDomain<Fr>::{fft, ifft, coset_fft, coset_ifft}_in_place (algebra/poly/src/domain/mixed_radix.rs:130-157,
 * 286-404): domains of `size` = 2^a or 3 * 2^a (SMALL_SUBGROUP_BASE = 3 with adicity 1, fr.rs:19-20) -- the Plonk prover's wire
 * domain has 3 * n_gates elements (mpc-plonk/src/relations/flat.rs:282-300).  Same conventions as czk_ntt_fr (data: lanes x size
 * x 4 u64, natural order in and out, in_len <= size, tail taken as zero); root = get_root_of_unity(size)
 * (algebra/ff/src/fields/mod.rs:337-367).  CZK_ERR_SIZE when no such domain exists.  A power-of-two size is the radix-2 transform.
 * czk_mixed_domain_constants: the six constants of czk_domain_constants for MixedRadixEvaluationDomain::new(size). */
int czk_ntt_fr_mixed(czk_ctx* ctx, uint64_t* data, size_t size, size_t lanes, int kind, size_t in_len, int mem);
int czk_mixed_domain_constants(czk_ctx* ctx, size_t size, uint64_t* out24);

/* ---- element-wise Fr vector ops (the share-local pointwise steps of witness_map) ------------------ */
/* out[i] = a[i] op b[i], n elements of 4 u64; out may alias a or b.  r1cs_to_qap.rs:92 (plain product), :105-107 (sub). */
typedef enum { CZK_OP_ADD = 0, CZK_OP_SUB = 1, CZK_OP_MUL = 2 } czk_binop;
int czk_fr_vec_op(czk_ctx* ctx, int op, const uint64_t* a, const uint64_t* b, uint64_t* out, size_t n, int mem);
/* out[i] = a[i] * k  (k: one Montgomery Fr in the SAME memory space as a / out; domain/mod.rs:184-191).  In DEVICE memory the call only
 * enqueues and the kernel reads `k` in stream order: `k` must stay valid and unmodified until the context's stream has passed this call (the
 * same holds for czk_ntt_fr_to's `src`). */
int czk_fr_vec_scale(czk_ctx* ctx, const uint64_t* a, const uint64_t* k, uint64_t* out, size_t n, int mem);
/* out[i] = c * g^i for i < n (g, c: one Montgomery Fr each, HOST memory; c NULL = one): the table behind
 * EvaluationDomain::distribute_powers / distribute_powers_and_mul_by_const (algebra/poly/src/domain/mod.rs:93-106) for any g --
 * coset shifts other than the generator, and `shift(p, w)` of the Plonk prover (mpc-plonk/src/util.rs).  Multiply with
 * czk_fr_vec_op(CZK_OP_MUL) per lane. */
int czk_fr_powers(czk_ctx* ctx, const uint64_t* g, const uint64_t* c, size_t n, uint64_t* out, int mem);
/* Local half of Beaver multiplication (mpc-algebra/src/share/field.rs:97-127), per lane:
 *   out[i] = z[i] - y[i]*sx[i] - x[i]*oy[i] + (add_open ? sx[i]*oy[i] : 0)
 * x, y, z: this party's triple shares; sx, oy: the opened values; add_open = this party applies `shift`
 * (king for the sh lane, mac_share != 0 for the mac lane; share/spdz.rs:31-37,204-208). */
int czk_fr_beaver_combine(czk_ctx* ctx, const uint64_t* x, const uint64_t* y, const uint64_t* z, const uint64_t* sx,
                          const uint64_t* oy, int add_open, uint64_t* out, size_t n, int mem);
/* Local arithmetic of SpdzFieldShare::batch_open (mpc-algebra/src/share/spdz.rs:166-185) once every party's shares
 * are on this GPU (parties as lanes, or after the all-gather of mpc-net's broadcast):
 *   shares: parties x 2 x n Fr (party-major; lane 0 = sh, lane 1 = mac), DEVICE memory
 *   value[i]  = sum_p sh_p[i]                                             -> out_value (n Fr, device)
 *   check[i]  = sum_p (mac_share_p * value[i] - mac_p[i]),  mac_share_p = (p == 0)   (spdz.rs:31-37, :176-183)
 * *out_bad (host) receives the number of i with check[i] != 0 (the reference asserts it is 0). */
int czk_fr_spdz_open(czk_ctx* ctx, const uint64_t* shares, size_t parties, size_t n, uint64_t* out_value, uint64_t* out_bad);

/* The same open round by round, as the reference runs it when every party is its own process (share/spdz.rs:166-185) --
 * between the calls the caller moves the vectors with mpc-net's broadcast / atomic_broadcast (RCCL all-gather here):
 *   round 1: every party broadcasts its `sh` lane;  values = czk_fr_lanes_sum(gathered sh lanes)
 *   local  : dx_t = czk_fr_spdz_dx(values, own mac lane, own mac_share)      (mac_share: one Montgomery Fr, HOST memory)
 *   round 2: atomic_broadcast(dx_t);  czk_fr_lanes_sum(gathered dx_t, out = NULL, &nonzero) must report 0.
 * czk_fr_lanes_sum: out[i] = sum_{j<k} x[j][i] over k vectors of n Fr (x: k x n, DEVICE); out (device, may be NULL) receives
 * the sums, *out_nonzero (host, may be NULL) the number of non-zero sums. */
int czk_fr_lanes_sum(czk_ctx* ctx, const uint64_t* x, size_t k, size_t n, uint64_t* out, uint64_t* out_nonzero);
int czk_fr_spdz_dx(czk_ctx* ctx, const uint64_t* value, const uint64_t* mac, const uint64_t* mac_share, uint64_t* out, size_t n);

/* GSZ / Shamir shares (mpc-algebra/src/share/gsz20/mod.rs): party j of n holds p(w^j), w = the order-n root of
 * MixedRadixEvaluationDomain::new(n_parties) (gsz20/mod.rs:98-105; algebra/poly/src/domain/mixed_radix.rs:57-107; root rule
 * algebra/ff/src/fields/mod.rs:337-367 with SMALL_SUBGROUP_BASE = 3).  A GSZ share vector is ONE Fr lane for the NTT / MSM
 * entry points (add / scale are lane-wise, gsz20/mod.rs:260-284).
 * czk_share_domain_constants: out[0..4) size_inv, [4..8) group_gen, [8..12) group_gen_inv (Montgomery limbs); CZK_ERR_SIZE when
 * n_parties is not 2^a or 3 * 2^a (the reference's `domain()` panics).
 * czk_fr_gsz_open = the local part of batch_open (:286-300) -> open_degree_vec (:440-466) once all parties' values are on this
 * GPU: shares: parties x n Fr (party-major, DEVICE); per element the size-n inverse DFT, p(0) -> out_value (n Fr, device);
 * *out_bad (host) = number of elements whose interpolating polynomial exceeds its degree bound (the reference asserts
 * p.degree() <= d): bound = degrees[i] (device u32 array) or `degree` when degrees is NULL. */
int czk_share_domain_constants(czk_ctx* ctx, size_t parties, uint64_t* out12);
int czk_fr_gsz_open(czk_ctx* ctx, const uint64_t* shares, size_t parties, size_t n, const uint32_t* degrees, unsigned degree,
                    uint64_t* out_value, uint64_t* out_bad);

/* ---- mpc-net between GPUs: the opens' transport, behind the C ABI -------------------------------- */
/* The reference's parties talk through the process-global MpcMultiNet (mpc-net/src/multi.rs:15-23: one TCP connection per pair of
 * parties) with four primitives -- broadcast (:145-173), send_to_king (:175-210), recv_from_king (:211-242) and, on top of them,
 * MpcSerNet::atomic_broadcast (mpc-algebra/src/channel.rs:50-75).  A czk_net is the same thing for parties that are czk contexts:
 * one communicator per (context, party), its exchanges enqueued on the context's stream (czk_ctx_stream), so a share vector that
 * the context's kernels produce is exchanged, summed and checked WITHOUT leaving HBM and without a host synchronisation between
 * the kernels and the exchange.  Two transports:
 *   CZK_NET_RCCL  one party per GPU of one node: RCCL over xGMI (ncclCommInitRank).  id = the 128 bytes czk_net_unique_id returned on
 *                 rank 0, carried to the other parties by whatever channel started them (the reference's host has mpc-net's TCP for it).
 *                 librccl.so.1 is dlopen-ed on first use: the library has no link-time dependency on RCCL.
 *   CZK_NET_SHM   parties are processes of one node in ANY assignment to GPUs -- several parties on ONE GPU included -- staged through a
 *                 POSIX shared-memory segment named by the id bytes (pinned by every party: two DMA copies per buffer).  For rigs with
 *                 fewer GPUs than parties and for tests; blocks the host for the duration of each exchange.  id: 1..32 arbitrary bytes
 *                 the launcher chooses, the same on every rank and FRESH for every communicator (czk_net_unique_id(CZK_NET_SHM) draws 16 random
 *                 bytes: a rank that finds the segment of an earlier run under its id would wait there until the timeout).  ctx may be NULL:
 *                 host-memory primitives only (no composite opens).  A failure on one rank aborts the communicator for all of them.
 *   CZK_NET_IPC   like CZK_NET_SHM (same id rule, same control block and barrier in shared memory), but the staging slots are DEVICE memory:
 *                 every rank owns a mailbox on its GPU and maps the others' with hipIpc, so an exchange is device-to-device copies (on one GPU: inside
 *                 HBM; across GPUs of a node: peer copies) and only the barrier touches the host.  Needs HSA_ENABLE_IPC_MODE_LEGACY=0 in the environment
 *                 of this driver stack; a context is required.
 * czk_net_create is collective (returns once every rank has joined; CZK_ERR_NET after the timeout).  All ranks call the same
 * sequence of exchanges, like the reference's lock-step rounds.  Buffers: `mem` = CZK_MEM_DEVICE (on the context's GPU, used in
 * stream order) or CZK_MEM_HOST (read / written before the call returns).  Byte counts must be equal on all ranks (mpc-net asserts it).
 * Options (czk_net_set_option): "exchange" 0 = ring (one ncclAllGather), 1 = p2p (world - 1 grouped ncclSend / ncclRecv pairs: every
 * pair of GPUs of an MI355X node has its own xGMI link -- mpc-net's own star shape); "timeout_ms" (default 120000);
 * "slot_bytes" (SHM: staging slot per rank, default 16 MiB; before the first exchange). */
typedef struct czk_net czk_net;
typedef enum { CZK_NET_RCCL = 1, CZK_NET_SHM = 2, CZK_NET_IPC = 3 } czk_net_transport;
#define CZK_NET_UNIQUE_ID_BYTES 128
int czk_net_unique_id(int transport, uint8_t* out, size_t cap, size_t* len);
int czk_net_create(czk_ctx* ctx, int transport, int rank, int world, const uint8_t* id, size_t id_len, czk_net** out);
void czk_net_destroy(czk_net* net);
int czk_net_rank(const czk_net* net);    /* MpcNet::party_id */
int czk_net_world(const czk_net* net);   /* MpcNet::n_parties */
int czk_net_set_option(czk_net* net, const char* name, long value);
const char* czk_net_last_error(const czk_net* net);
/* mpc-net's Stats (mpc-net/src/lib.rs: bytes_sent, bytes_recv, broadcasts, to_king, from_king), counted by the reference's rules:
 * out[0..5) in that order.  czk_net_stats_reset = MpcNet::reset_stats. */
int czk_net_stats(const czk_net* net, uint64_t* out5);
void czk_net_stats_reset(czk_net* net);
/* broadcast (multi.rs:145-173): every party contributes `bytes` bytes; recv (world x bytes, party order) receives all of them. */
int czk_net_broadcast(czk_net* net, const void* send, size_t bytes, void* recv, int mem);
/* send_to_king (multi.rs:175-210): recv (world x bytes, party order) is written on the king (rank 0) only and may be NULL elsewhere. */
int czk_net_send_to_king(czk_net* net, const void* send, size_t bytes, void* recv, int mem);
/* recv_from_king (multi.rs:211-242): the king hands party p the p-th of `world` equally long buffers (send: world x bytes on the
 * king, ignored elsewhere); recv (bytes) receives this party's.  The reference prefixes each message with its u64 length
 * (multi.rs:219-228); here lengths are arguments and only the stats count the 8 bytes. */
int czk_net_recv_from_king(czk_net* net, const void* send, size_t bytes, void* recv, int mem);
int czk_net_barrier(czk_net* net);
/* MpcSerNet::atomic_broadcast of one Vec<Fr> (channel.rs:50-75): round 1 broadcasts SHA-256(serialize(x) || 32 random bytes), round 2
 * the vector and the randomness; every receiver re-hashes what the others sent (CZK_ERR_CHECK on a mismatch).  x: n Montgomery Fr
 * (`mem`), recv: world x n Fr.  The commitment runs over the reference's wire bytes (czk_fr_vec_serialize), so equal data and
 * randomness give the reference's hashes.  rand32: the 32 commitment bytes (tests), NULL = drawn from the OS.  Hashing is host
 * work over a download of the vectors: the protocol's check, not hot-path arithmetic -- the opens below take it as an option. */
int czk_net_atomic_broadcast(czk_net* net, const uint64_t* x, size_t n, uint64_t* recv, const uint8_t* rand32, int mem);

/* The reference's batch opens as ONE call each on device lanes (all buffers CZK_MEM_DEVICE on the communicator's context; gathered
 * shares live in the communicator's own scratch).  flags: CZK_OPEN_COMMIT = the second round goes through czk_net_atomic_broadcast.
 *   czk_spdz_batch_open  SpdzFieldShare::batch_open (mpc-algebra/src/share/spdz.rs:166-185): broadcast of the sh lane, value = sum;
 *                        dx_t = mac_share * value - mac; (atomic_)broadcast of dx_t; *out_bad = number of i whose dx_t do not sum to
 *                        zero (the reference asserts 0).  sh, mac: n Fr each; mac_share: one Montgomery Fr, HOST (1 on the king, 0
 *                        elsewhere with the reference's stand-in key, spdz.rs:30-37); MAC shares never leave the party.
 *   czk_add_batch_open   AdditiveFieldShare::batch_open (share/add.rs:256-259): broadcast, sum.
 *   czk_gsz_batch_open   GszFieldShare::batch_open (share/gsz20/mod.rs:286-300) -> open_degree_vec (:440-466): broadcast, per element
 *                        the size-world inverse DFT, degree check, p(0); degrees / degree / out_bad as in czk_fr_gsz_open.
 * out_value may alias the input share lane.  Each call ends with the read-back of its check count (the reference asserts right there). */
#define CZK_OPEN_COMMIT 1
int czk_spdz_batch_open(czk_net* net, const uint64_t* sh, const uint64_t* mac, const uint64_t* mac_share, size_t n, uint64_t* out_value,
                        int flags, uint64_t* out_bad);
int czk_add_batch_open(czk_net* net, const uint64_t* val, size_t n, uint64_t* out_value);
int czk_gsz_batch_open(czk_net* net, const uint64_t* val, size_t n, const uint32_t* degrees, unsigned degree, uint64_t* out_value,
                       uint64_t* out_bad);
/* The king's side of king_compute (mpc-algebra/src/channel.rs:77-80; gsz20's degree reduction, share/gsz20/mod.rs:470-500): every
 * party's n Fr to the king, who receives world x n (device, party order) -- czk_net_send_to_king on Fr lanes -- and the way back. */
int czk_fr_send_to_king(czk_net* net, const uint64_t* x, size_t n, uint64_t* gathered);
int czk_fr_recv_from_king(czk_net* net, const uint64_t* parts, size_t n, uint64_t* out);
/* gsz20::batch_king_compute with f = the identity -- the only f the reference passes (share/gsz20/mod.rs:494-527, called at :547,
 * :578, :805: "king just reduces the sharing degree" of a product share x * y + r2): every party's lane to the king, the king opens
 * each element with the degree bound (czk_fr_gsz_open) and sends the opened VALUE back to every party as its new share (:478
 * `vec![output; n]`, "TODO: randomize").  val: n Fr (device); out (n Fr, device; may alias val) = from_king; *out_bad is meaningful
 * on the king (0 elsewhere). */
int czk_gsz_batch_king_compute(czk_net* net, const uint64_t* val, size_t n, const uint32_t* degrees, unsigned degree, uint64_t* out,
                               uint64_t* out_bad);

/* `Vec<Fr>::serialize` / `deserialize` (algebra/serialize/src/lib.rs:220-229 with impl_prime_field_serializer, fields/macros.rs:
 * 532-537): u64 little-endian length, then into_repr() of every element as 32 little-endian bytes.  For a host that still carries
 * some vectors over the reference's own sockets.  serialize: a = n Montgomery Fr (`mem`), out = 8 + 32 n bytes, HOST.  deserialize:
 * bytes / len as received; writes at most cap Fr (Montgomery) to out (`mem`), *n = the length prefix; CZK_ERR_ARG when the prefix
 * does not match len (the reference's deserialize fails) or exceeds cap. */
int czk_fr_vec_serialize(czk_ctx* ctx, const uint64_t* a, size_t n, int mem, uint8_t* out);
int czk_fr_vec_deserialize(czk_ctx* ctx, const uint8_t* bytes, size_t len, uint64_t* out, size_t cap, int mem, size_t* n);
/* SHA-256 (FIPS 180-4) of `len` host bytes: the reference's CommitHash (channel.rs:92).  Exported so a host can reproduce / verify
 * commitments; no context needed. */
void czk_sha256(const void* data, size_t len, uint8_t* out32);

/* ---- callers either side of the NTT: constraint evaluation and division by (X - z) ------------------ */
/* R1CS matrix (one of ConstraintMatrices::{a, b, c}, Vec<Vec<(F, usize)>>) in CSR form, pinned on the GPU once per
 * circuit: row_ptr m+1 offsets (row_ptr[0] = 0, row_ptr[m] = nnz), col_idx nnz variable indices into the full
 * assignment [instance | witness] (r1cs_to_qap.rs:56-61), coeff nnz Montgomery Fr.  CZK_ERR_ARG for a malformed
 * row_ptr or an index >= n_vars (the reference panics on assignment[index]). */
typedef struct czk_r1cs_matrix czk_r1cs_matrix;
int czk_r1cs_matrix_register(czk_ctx* ctx, const uint64_t* row_ptr, const uint32_t* col_idx, const uint64_t* coeff, size_t m,
                             size_t nnz, size_t n_vars, int mem, czk_r1cs_matrix** out);
void czk_r1cs_matrix_release(czk_r1cs_matrix* a);
/* evaluate_constraint over every row (mpc-snarks/src/groth/r1cs_to_qap.rs:12-42, called at :70-77 and :95-100):
 *   out[lane][i] = sum_t coeff[t] * z[lane][col_idx[t]],  i < m
 * z: lanes x z_stride Fr (z_stride >= n_vars), the share lanes of the full assignment with Public entries lifted as for
 * the NTT (wire/field.rs Public * coeff stays public; lifting commutes with the sum); out: lanes x out_stride Fr
 * (out_stride >= m; elements [m, out_stride) are not written, so the caller's zero padding to the domain size stays). */
int czk_r1cs_matvec(czk_ctx* ctx, const czk_r1cs_matrix* a, const uint64_t* z, size_t z_stride, size_t lanes, uint64_t* out,
                    size_t out_stride, int mem);
/* DensePolynomial / (X - z): KZG10::compute_witness_polynomial (poly-commit/src/kzg10/mod.rs:200-224; shares divide
 * lane-wise because the divisor is public, mpc-algebra/src/share/add.rs:148-156).  coeffs: lanes x n Fr, low degree
 * first; quotient: lanes x (n-1) Fr; remainder (may be NULL): lanes Fr = p(z), which is also
 * Polynomial::evaluate (kzg10/mod.rs:247).  z: one Montgomery Fr in HOST memory. */
int czk_poly_div_linear(czk_ctx* ctx, const uint64_t* coeffs, size_t n, size_t lanes, const uint64_t* z, uint64_t* quotient,
                        uint64_t* remainder, int mem);
/* DensePolynomial::evaluate (algebra/poly/src/polynomial/univariate/dense.rs:59-96, Horner) per lane, without the quotient: the
 * evaluations a prover publishes at its challenge points (mpc-plonk/src/lib.rs:273, :360; poly-commit/src/kzg10/mod.rs:246, :525;
 * Marlin's evaluation round).  coeffs: lanes x n Fr, low degree first; values: lanes Fr = p(z); z: one Montgomery Fr in HOST
 * memory.  Same result as czk_poly_div_linear's remainder, at half its memory traffic. */
int czk_poly_evaluate(czk_ctx* ctx, const uint64_t* coeffs, size_t n, size_t lanes, const uint64_t* z, uint64_t* values, int mem);
/* DensePolynomial::divide_by_vanishing_poly (algebra/poly/src/polynomial/univariate/dense.rs:172-179) for the radix-2 / mixed-radix vanishing polynomial
 * X^n - 1, per lane (the divisor is public): a = q (X^n - 1) + r.  coeffs: lanes x m Fr, low degree first, m >= n; quotient: lanes x (m - n) Fr;
 * remainder (may be NULL): lanes x n Fr.  The quotients h of mpc-plonk's gate / product arguments (mpc-plonk/src/lib.rs:190, :332) and Marlin's h_1, h_2
 * (marlin/src/ahp/prover.rs:533, :696); one pass over the coefficients.  (For n of a few units -- Marlin's v_X with |X| = 2 -- the residue classes are long
 * chains: divide them with czk_poly_div_linear at z = 1 instead, as tools/polyvm_host.hpp does.) */
/* czk_poly_evaluate for `count` polynomials in one call, DEVICE memory: polynomial k = n[k] coefficients on lanes[k] lanes at coeffs[k], evaluated at
 * z + 4 k (HOST, Montgomery), its lanes[k] values written to values[k].  One launch per 32-fold reduction level for all of them: Marlin's evaluation round
 * (marlin/src/lib.rs:283-292 through EvaluationsProvider::get_lc_eval, marlin/src/ahp/mod.rs:288-312) evaluates two dozen polynomials at two points,
 * and the upper levels of one evaluation are a few dozen elements -- launch-bound when issued one polynomial at a time. */
int czk_poly_evaluate_many(czk_ctx* ctx, size_t count, const uint64_t* const* coeffs, const size_t* n, const size_t* lanes, const uint64_t* z,
                           uint64_t* const* values);
/* out[l][i] = constant + sum_k coeffs[k] * terms[k][l][i], i < out_len, over `count` <= 12 DEVICE arrays of different lengths (term_len[k] elements per lane; shorter
 * terms end early, longer ones are cut at out_len), coeffs: count x 4 u64, HOST, Montgomery; constant: one Fr, HOST, Montgomery, or NULL (zero).  A term has
 * `lanes` lanes (term_lanes[k] == lanes) or is PUBLIC (term_lanes[k] == 1): a public term -- and the constant, which is public -- is added on the lanes whose
 * bit is set in `lift_mask` only -- the rule by which the reference adds a public
 * value to a shared one (AdditiveFieldShare::shift, mpc-algebra/src/share/add.rs:141-146: the king; GszFieldShare: every party).  One pass: the linear
 * combinations of polynomials a prover forms -- `poly += (*coeff, cur_poly.polynomial())` per term (poly-commit/src/marlin/mod.rs:275; the batch opening's
 * fold, marlin_pc/mod.rs:259-316; marlin/src/ahp/prover.rs:468-476) -- where a scale / resize / add chain makes a pass over memory per call. */
int czk_fr_lincomb(czk_ctx* ctx, size_t count, const uint64_t* const* terms, const size_t* term_len, const size_t* term_lanes, const uint64_t* coeffs,
                   const uint64_t* constant, size_t lanes, uint64_t lift_mask, uint64_t* out, size_t out_len);
int czk_poly_div_vanishing(czk_ctx* ctx, const uint64_t* coeffs, size_t m, size_t lanes, size_t n, uint64_t* quotient, uint64_t* remainder, int mem);

/* out[i] = x[0] * x[1] * ... * x[i] over a PUBLIC vector: the sequential loop of partial_products between its
 * batch_open and the final scale (mpc-algebra/src/share/field.rs:169-172; Plonk's grand product).  The share-side
 * scale that follows (:177-179) is czk_fr_vec_op(CZK_OP_MUL) per lane.  out may alias x only in HOST mode. */
int czk_fr_prefix_product(czk_ctx* ctx, const uint64_t* x, size_t n, uint64_t* out, int mem);

/* batch_inversion_and_mul (algebra/ff/src/fields/mod.rs:616-677): out[i] = coeff * v[i]^-1, zero elements stay zero
 * (the reference skips them, :651, :666).  coeff: one Montgomery Fr in HOST memory, NULL = one (batch_inversion, :616).
 * In device memory out must not alias v. */
int czk_fr_batch_inverse(czk_ctx* ctx, const uint64_t* v, size_t n, const uint64_t* coeff, uint64_t* out, int mem);

/* Fr::into_repr / from_repr over a vector (fields/arithmetic.rs:59-81, macros.rs:443-454) -- also the wire format. */
int czk_fr_into_repr(czk_ctx* ctx, const uint64_t* a, uint64_t* out, size_t n, int mem);
int czk_fr_from_repr(czk_ctx* ctx, const uint64_t* a, uint64_t* out, size_t n, int mem);

/* ---- MSM ---------------------------------------------------------------------------------------- */
/* Pin a public base array (a proving-key query) on the GPU once; it is reused across proofs
 * (groth16/src/data_structures.rs:132-149).  bases: n x (12|24) u64 affine Montgomery; inf: n bytes (may be NULL =
 * no infinity points). */
int czk_bases_register(czk_ctx* ctx, int group, const uint64_t* bases, const uint8_t* inf, size_t n, int mem,
                       czk_bases** out);
void czk_bases_release(czk_bases* b);
size_t czk_bases_len(const czk_bases* b);
/* Pippenger layout chosen at registration (reporting only): *c = signed-digit window width, *windows = ceil(254 / c) =
 * mixed additions per (point, lane) of an MSM over this array.  Where c * windows overshoots the 254 bits by `slack` bits and the top window
 * would be narrower than 10 bits (widths chosen for fewer than 2^14 points), the last `slack` windows are c - 1 bits wide instead, so that
 * no window is narrow and no bucket collects more than twice the average (csrc/czk_internal.h: msm_full_windows). */
int czk_bases_layout(const czk_bases* b, unsigned* c, unsigned* windows);
/* Window width per CALL.  The reference derives c from the size of each MSM (variable_base.rs:21-25); a precomputed table fixes
 * it per key, so a short MSM under a long key (KZG10::commit of a low-degree polynomial under `powers_of_g`,
 * poly-commit/src/kzg10/mod.rs:159-162) would reduce the key's 2^(c-1) buckets on every call.  MSMs that use a short prefix of
 * the array therefore run on a second table set at a narrower width (c = 13 / 15 / 17 by size, covering the next power of two of
 * the call), built on first use and kept with the handle; czk_bases_prepare builds the set for calls of `n_scalars` scalars up
 * front (e.g. at SRS load), czk_bases_layout_for reports the (c, windows) such a call runs with.  Handles registered with
 * CZK_MEM_NO_TABLES choose c per call outright. */
int czk_bases_layout_for(const czk_bases* b, size_t n_scalars, unsigned* c, unsigned* windows);
/* Bucket arithmetic the handle's MSMs run with (reporting only): 0 = XYZZ, saturated limbs; 1 = XYZZ, unsaturated limbs (8M + 2S per mixed
 * addition); 2 = twisted Edwards extended coordinates, unsaturated limbs (G1 in the prime-order subgroup: 7M per mixed addition). */
int czk_bases_arith(const czk_bases* b);
int czk_bases_prepare(czk_ctx* ctx, const czk_bases* b, size_t n_scalars);

/* Replaces VariableBaseMSM::multi_scalar_mul (algebra/ec/src/msm/variable_base.rs:12-106) / AffineCurve::
 * multi_scalar_mul (ec/src/lib.rs:300-311) as reached from MpcG{1,2}Affine::multi_scalar_mul
 * (mpc-algebra/src/wire/pairing.rs:746-809) -> GroupShare::multi_scale_pub_group (share/spdz.rs:440-446).
 * scalars: lanes x n_scalars x 4 u64 (lane-major); `lanes` scalar vectors share the bases (SPDZ: sh and mac).
 * size = min(czk_bases_len, n_scalars) pairs are used (variable_base.rs:16; h has D scalars vs D-1 bases).
 * out_jac: lanes x (18|36) u64 Jacobian, Montgomery, HOST memory.  Jacobian triples are not canonical: compare in
 * affine (czk_jac_to_affine), as the reference's own MSM test does (algebra/test-templates/src/msm.rs:16-33). */
int czk_msm(czk_ctx* ctx, const czk_bases* bases, const uint64_t* scalars, size_t n_scalars, size_t lanes,
            int scalar_form, int mem, uint64_t* out_jac);

/* Same, but only enqueues: consecutive calls overlap on the context's internal streams (sort / accumulate / reduce
 * stages of neighbouring MSMs run concurrently).  `out_jac` (host) is valid after the next czk_ctx_sync(), or after czk_ctx_wait_mark on a
 * mark taken after this call.  Device
 * scalars may be overwritten by later work on the context's stream (the library orders that itself). */
int czk_msm_async(czk_ctx* ctx, const czk_bases* bases, const uint64_t* scalars, size_t n_scalars, size_t lanes,
                  int scalar_form, int mem, uint64_t* out_jac);

/* One-shot forms with the reference's argument order (bases not kept on the GPU).  These are VariableBaseMSM::multi_scalar_mul's
 * signature, which is complete on every curve point, so they make NO subgroup assumption: the bases are registered with
 * CZK_MEM_NO_TABLES | CZK_MEM_ANY_POINTS for the call (G1: XYZZ bucket arithmetic with the reference's case analysis). */
int czk_msm_g1(czk_ctx* ctx, const uint64_t* bases_xy, const uint8_t* inf, const uint64_t* scalars, size_t n,
               size_t lanes, int scalar_form, uint64_t* out_jac);
int czk_msm_g2(czk_ctx* ctx, const uint64_t* bases_xy, const uint8_t* inf, const uint64_t* scalars, size_t n,
               size_t lanes, int scalar_form, uint64_t* out_jac);

/* GroupAffine::is_in_correct_subgroup_assuming_on_curve (short_weierstrass_jacobian.rs:131: `self.mul(r).is_zero()`) over the
 * registered bases, on the GPU: *out_bad = number of bases that are not on the curve or not annihilated by r (points at infinity
 * pass).  0 means the default (twisted Edwards) G1 arithmetic is exact for this handle.  Blocking.  For a handle registered with
 * CZK_MEM_CHECK_SUBGROUP the stored result of the registration-time check is returned. */
int czk_bases_check_subgroup(czk_ctx* ctx, const czk_bases* bases, size_t* out_bad);

/* From<GroupProjective> for GroupAffine (short_weierstrass_jacobian.rs:768-789): n host Jacobian points ->
 * n host affine points + infinity flags.  This is what AffineMsm::msm's `.into()` does (share/msm.rs:31-37).
 * Host-side arithmetic (one inversion per point); `ctx` may be NULL. */
int czk_jac_to_affine(czk_ctx* ctx, int group, const uint64_t* jac, size_t n, uint64_t* out_aff, uint8_t* out_inf);

/* GroupProjective += (add_assign, short_weierstrass_jacobian.rs:666-728) and += affine (add_assign_mixed, :570-638) on
 * host Jacobian values: the O(1) group steps around the MSMs (calculate_coeff, prover.rs:216-232; KZG10::commit's
 * `commitment.add_assign_mixed(&random_commitment)`, poly-commit/src/kzg10/mod.rs:188).  Host arithmetic; ctx may be NULL. */
int czk_jac_add(czk_ctx* ctx, int group, const uint64_t* a_jac, const uint64_t* b_jac, uint64_t* out_jac);
int czk_jac_add_mixed(czk_ctx* ctx, int group, const uint64_t* a_jac, const uint64_t* b_aff, int b_inf, uint64_t* out_jac);
/* ProjectiveCurve::mul / scalar_mul (algebra/ec/src/lib.rs:215-230: double-and-add over the scalar's bits, most significant first) and Neg
 * (short_weierstrass_jacobian.rs:737-748: (x, -y, z)) on host Jacobian values: `pk.delta_g1.scalar_mul(r)`, `g_a.scalar_mul(&s)`,
 * `g_c -= &r_s_delta_g1` of create_proof (mpc-snarks/src/groth/prover.rs:113-165).  k: one Fr, HOST memory, `scalar_form` as for the MSM.
 * Host arithmetic (about a millisecond per G1 multiplication); ctx may be NULL. */
int czk_jac_scalar_mul(czk_ctx* ctx, int group, const uint64_t* a_jac, const uint64_t* k, int scalar_form, uint64_t* out_jac);
int czk_jac_neg(czk_ctx* ctx, int group, const uint64_t* a_jac, uint64_t* out_jac);

/* Synthetic public bases P_i = [k_i] * generator for i < n, k_i = canonical scalars (n x 4 u64), written as
 * affine Montgomery points to `out` (device or host).  Stands in for a trusted-setup run when benchmarking
 * (SURVEY.md section 8d); also how tests obtain on-curve points with known discrete logs. */
int czk_fixed_base_points(czk_ctx* ctx, int group, const uint64_t* k, size_t n, uint64_t* out, int mem);

/* ---- Groth16 per-party local compute (callers of the two kernels) -------------------------------- */
/* R1CStoQAP::witness_map minus its communication step (mpc-snarks/src/groth/r1cs_to_qap.rs:47-113), on `lanes`
 * Fr lanes of D = 2^log_d elements, all buffers lanes x D x 4 u64 in DEVICE memory:
 *   czk_witness_map_pre : a <- coset_fft(ifft(a)), b <- coset_fft(ifft(b))                       (:85-89)
 *   [caller: ab = batch_product(a, b) -- Beaver opens for shares, czk_fr_vec_op(MUL) for a single prover] (:92)
 *   czk_witness_map_post: c <- coset_fft(ifft(c)); ab <- coset_ifft((ab - c) * Z(g)^-1)          (:102-110)
 * h = ab on return.  a_len / b_len / c_len <= D: evaluations present in each lane (num_constraints + num_inputs for a,
 * num_constraints for b and c); elements beyond them are the reference's `vec![zero; domain_size]` padding (:66-67, :94)
 * and are never read -- the first NTT pass zero-extends. */
int czk_witness_map_pre(czk_ctx* ctx, uint64_t* a, size_t a_len, uint64_t* b, size_t b_len, unsigned log_d, size_t lanes);
int czk_witness_map_post(czk_ctx* ctx, uint64_t* ab, uint64_t* c, size_t c_len, unsigned log_d, size_t lanes);

/* ---- measurement hooks --------------------------------------------------------------------------- */
/* When enabled, the library brackets its kernel launches with HIP events on the context's stream (the stream the
 * kernels run on) and accumulates per-kernel elapsed time.  Names: "msm_accumulate_g1", "msm_accumulate_g2",
 * "msm_sort", "msm_reduce", "ntt_pass", "pointwise".  czk_profile_read synchronises the stream. */
int czk_profile_enable(czk_ctx* ctx, int on);
int czk_profile_reset(czk_ctx* ctx);
int czk_profile_read(czk_ctx* ctx, const char* kernel, double* total_ms, uint64_t* launches);
/* The same brackets as intervals: (start, stop) of every bracket of `kernel` in milliseconds after the context's profile origin (an
 * event czk_profile_reset records on the context's stream).  Writes at most `cap` pairs, *n = the number available.  Summed elapsed
 * times of brackets on different streams overlap; the UNION of the intervals is the time a kernel of that name was running -- what
 * bench.py reports as accumulate-busy time.  czk_profile_base_offset: origin of `b` minus origin of `a` (same device), to merge the
 * intervals of several contexts onto one clock. */
int czk_profile_intervals(czk_ctx* ctx, const char* kernel, double* start_ms, double* stop_ms, size_t cap, size_t* n);
int czk_profile_base_offset(czk_ctx* a, czk_ctx* b, double* ms);

#ifdef __cplusplus
}
#endif
#endif /* CZK_H */
