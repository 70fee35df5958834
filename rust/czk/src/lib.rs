//! Safe layer over `czk-sys` for the three seams of alex-ozdemir/collaborative-zksnark.
//!
//! * NTT seam  -- `EvaluationDomain::{fft, ifft, coset_fft, coset_ifft}_in_place`
//!   (algebra/poly/src/domain/radix2/mod.rs:99-117, domain/mod.rs:139-142): [`ntt::transform_in_place`]
//! * MSM seam  -- `AffineCurve::multi_scalar_mul` / `VariableBaseMSM::multi_scalar_mul`
//!   (algebra/ec/src/lib.rs:300-311, algebra/ec/src/msm/variable_base.rs:12-106): [`msm::g1`], [`msm::g2`]
//! * share seam -- `MpcField::{Public, Shared}` vectors as Fr lanes (mpc-algebra/src/wire/field.rs:27-30, 89-105;
//!   share/spdz.rs:31-37, 186-208): [`share::SpdzLanes`]
//!
//! Error convention: the reference panics (`assert!`, `unwrap()`, `None.unwrap()`); every wrapper here `expect()`s the C
//! status with the library's error text, so behaviour at the seams is unchanged.
//!
//! This crate cannot be compiled in the build image of this repository (no Rust toolchain); `include/czk.hpp` +
//! `tools/host_demo.cpp` are the compiled and GPU-tested C++ equivalents of the same logic, line for line.

use std::ffi::CStr;
use std::os::raw::c_int;
use std::ptr;

use czk_sys as sys;

/// One GPU + one HIP stream = one MPC party (`mpc-net/src/multi.rs:15-23`: one party per process, calls strictly sequential).
pub struct Context {
    raw: *mut sys::czk_ctx,
}

// The reference prover is single-threaded; the context is only ever used behind the global mutex below.
unsafe impl Send for Context {}

impl Context {
    /// `device`: the GPU of this party (`party id % visible GPUs` in the multi-process layout).
    pub fn new(device: i32) -> Context {
        let mut raw: *mut sys::czk_ctx = ptr::null_mut();
        let rc = unsafe { sys::czk_ctx_create(&mut raw, device as c_int, ptr::null_mut()) };
        if rc != sys::CZK_OK || raw.is_null() {
            panic!("czk_ctx_create failed with status {} (no visible MI355X? there is no CPU fallback)", rc);
        }
        Context { raw }
    }

    pub fn as_ptr(&self) -> *mut sys::czk_ctx {
        self.raw
    }

    /// Panics with the library's message unless `rc == CZK_OK` -- the shim's `expect()`.
    pub fn expect(&self, rc: c_int, what: &str) {
        if rc != sys::CZK_OK {
            let msg = unsafe { CStr::from_ptr(sys::czk_last_error(self.raw)) }
                .to_string_lossy()
                .into_owned();
            panic!("{}: czk status {}: {}", what, rc, msg);
        }
    }

    pub fn sync(&self) {
        let rc = unsafe { sys::czk_ctx_sync(self.raw) };
        self.expect(rc, "czk_ctx_sync");
    }
}

impl Drop for Context {
    fn drop(&mut self) {
        unsafe { sys::czk_ctx_destroy(self.raw) }
    }
}

lazy_static::lazy_static! {
    /// The party's context.  `CZK_DEVICE` selects the GPU (default 0); with one process per party on an 8-GPU node the
    /// launcher exports `CZK_DEVICE = party id`.
    pub static ref CTX: std::sync::Mutex<Context> = {
        let device = std::env::var("CZK_DEVICE").ok().and_then(|v| v.parse().ok()).unwrap_or(0);
        std::sync::Mutex::new(Context::new(device))
    };
}

pub mod limbs {
    //! Repacking between arkworks values and the library's limb arrays.  arkworks structs have no `#[repr(C)]`, so nothing
    //! is transmuted: every value is copied limb by limb.
    use ark_bls12_377::{Fq, Fq2, Fr, G1Affine, G1Projective, G2Affine, G2Projective};
    use ark_ff::{BigInteger256, BigInteger384};

    /// Montgomery limbs of an `Fr` (`Fp256(BigInteger256([u64; 4]), _)`, algebra/ff/src/fields/macros.rs:103-108).
    #[inline]
    pub fn fr_to(x: &Fr, out: &mut [u64]) {
        out[..4].copy_from_slice(&(x.0).0);
    }
    #[inline]
    pub fn fr_from(l: &[u64]) -> Fr {
        Fr::new(BigInteger256::new([l[0], l[1], l[2], l[3]]))
    }
    #[inline]
    pub fn fq_to(x: &Fq, out: &mut [u64]) {
        out[..6].copy_from_slice(&(x.0).0);
    }
    #[inline]
    pub fn fq_from(l: &[u64]) -> Fq {
        Fq::new(BigInteger384::new([l[0], l[1], l[2], l[3], l[4], l[5]]))
    }
    #[inline]
    pub fn fq2_to(x: &Fq2, out: &mut [u64]) {
        fq_to(&x.c0, &mut out[..6]);
        fq_to(&x.c1, &mut out[6..12]);
    }
    #[inline]
    pub fn fq2_from(l: &[u64]) -> Fq2 {
        Fq2::new(fq_from(&l[..6]), fq_from(&l[6..12]))
    }

    /// G1 affine points -> (n x 12 limbs, n infinity flags)  (`GroupAffine { x, y, infinity }`,
    /// algebra/ec/src/models/short_weierstrass_jacobian.rs:43-49)
    pub fn g1_bases(bases: &[G1Affine]) -> (Vec<u64>, Vec<u8>) {
        let mut xy = vec![0u64; 12 * bases.len()];
        let mut inf = vec![0u8; bases.len()];
        for (i, b) in bases.iter().enumerate() {
            fq_to(&b.x, &mut xy[12 * i..12 * i + 6]);
            fq_to(&b.y, &mut xy[12 * i + 6..12 * i + 12]);
            inf[i] = b.infinity as u8;
        }
        (xy, inf)
    }
    pub fn g2_bases(bases: &[G2Affine]) -> (Vec<u64>, Vec<u8>) {
        let mut xy = vec![0u64; 24 * bases.len()];
        let mut inf = vec![0u8; bases.len()];
        for (i, b) in bases.iter().enumerate() {
            fq2_to(&b.x, &mut xy[24 * i..24 * i + 12]);
            fq2_to(&b.y, &mut xy[24 * i + 12..24 * i + 24]);
            inf[i] = b.infinity as u8;
        }
        (xy, inf)
    }
    /// 18 limbs -> `GroupProjective::new(x, y, z)` (:420); z == 0 is the point at infinity (:444-457)
    pub fn g1_from_jac(l: &[u64]) -> G1Projective {
        G1Projective::new(fq_from(&l[..6]), fq_from(&l[6..12]), fq_from(&l[12..18]))
    }
    pub fn g2_from_jac(l: &[u64]) -> G2Projective {
        G2Projective::new(fq2_from(&l[..12]), fq2_from(&l[12..24]), fq2_from(&l[24..36]))
    }
}

pub mod ntt {
    //! `Radix2EvaluationDomain<Fr>::{fft, ifft, coset_fft, coset_ifft}_in_place` for element types that are a fixed number
    //! of `Fr` lanes (plain `Fr`: 1; `MpcField<Fr, SpdzFieldShare<Fr>>`: 2 -- see `share`).
    use super::{sys, CTX};
    use ark_bls12_377::Fr;

    #[derive(Clone, Copy, PartialEq, Eq, Debug)]
    pub enum Kind {
        Fft,
        Ifft,
        CosetFft,
        CosetIfft,
    }
    impl Kind {
        fn raw(self) -> std::os::raw::c_int {
            match self {
                Kind::Fft => sys::CZK_FFT,
                Kind::Ifft => sys::CZK_IFFT,
                Kind::CosetFft => sys::CZK_COSET_FFT,
                Kind::CosetIfft => sys::CZK_COSET_IFFT,
            }
        }
    }

    /// An element type the GPU can transform: `LANES` Fr values per element, stored lane-major (SoA) for the library.
    /// This is the hook `DomainCoeff` gains in the reference (see rust/PATCHES.md, patch 2).
    pub trait Lanes: Sized + Clone {
        const LANES: usize;
        /// Writes `v` (at most `d` elements) into `out` = `LANES` lanes of `d` elements of 4 limbs; elements beyond `v.len()`
        /// need not be written (the library takes them as zero: `resize(size, T::zero())`).
        fn pack(v: &[Self], d: usize, out: &mut [u64]);
        /// Rebuilds `d` elements from the transformed lanes.
        fn unpack(lanes: &[u64], d: usize) -> Vec<Self>;
    }

    impl Lanes for Fr {
        const LANES: usize = 1;
        fn pack(v: &[Fr], _d: usize, out: &mut [u64]) {
            for (i, x) in v.iter().enumerate() {
                super::limbs::fr_to(x, &mut out[4 * i..4 * i + 4]);
            }
        }
        fn unpack(lanes: &[u64], d: usize) -> Vec<Fr> {
            (0..d).map(|i| super::limbs::fr_from(&lanes[4 * i..4 * i + 4])).collect()
        }
    }

    /// The body of `fft_in_place` & co. with the GPU behind it.  `log_size_of_group` is the domain's
    /// (`Radix2EvaluationDomain::log_size_of_group`, radix2/mod.rs:27); the `assert!(coeffs.len() <= size)` of the reference
    /// (:100) is `CZK_ERR_SIZE` and panics through `expect`.
    pub fn transform_in_place<T: Lanes>(log_size_of_group: u32, coeffs: &mut Vec<T>, kind: Kind) {
        let d = 1usize << log_size_of_group;
        let in_len = coeffs.len();
        let mut lanes = vec![0u64; T::LANES * d * 4];
        T::pack(coeffs.as_slice(), d, &mut lanes);
        let ctx = CTX.lock().unwrap();
        let rc = unsafe {
            sys::czk_ntt_fr(ctx.as_ptr(), lanes.as_mut_ptr(), log_size_of_group, T::LANES, kind.raw(), in_len, sys::CZK_MEM_HOST)
        };
        ctx.expect(rc, "czk_ntt_fr");
        *coeffs = T::unpack(&lanes, d);
    }
}

pub mod msm {
    //! `AffineCurve::multi_scalar_mul(bases, scalars)` for BLS12-377 G1 / G2 (algebra/ec/src/lib.rs:300-311).  Scalars are
    //! passed in Montgomery form; `into_repr` (:305-307) runs on the GPU.  Lengths may differ: the library uses
    //! `min(bases.len(), scalars.len())` pairs like the reference (variable_base.rs:16).
    use super::{limbs, sys, CTX};
    use ark_bls12_377::{Fr, G1Affine, G1Projective, G2Affine, G2Projective};
    use std::collections::HashMap;
    use std::sync::Mutex;

    /// A registered base slice: the library handle, whether it carries window tables, and how often it has been used.
    struct Registered {
        handle: *mut sys::czk_bases,
        tables: bool,
        uses: u32,
    }
    unsafe impl Send for Registered {}

    lazy_static::lazy_static! {
        // Base slices are found again by (address, length, group).  The first MSM over a slice registers it WITHOUT window
        // tables (CZK_MEM_NO_TABLES: a copy, no precomputation -- right for the reference's prove-once binaries and for
        // one-off commitments); a slice that comes back (proving-key queries are reused across proofs,
        // groth16/src/data_structures.rs:132-149) is re-registered with tables, which makes every later MSM ~1.4x cheaper.
        static ref BASES: Mutex<HashMap<(usize, usize, i32), Registered>> = Mutex::new(HashMap::new());
    }

    fn scalars_to_limbs(scalars: &[Fr]) -> Vec<u64> {
        let mut s = vec![0u64; 4 * scalars.len()];
        for (i, x) in scalars.iter().enumerate() {
            limbs::fr_to(x, &mut s[4 * i..4 * i + 4]);
        }
        s
    }

    fn register(ctx: &super::Context, group: i32, pts: &[u64], inf: &[u8], tables: bool) -> *mut sys::czk_bases {
        let mut h: *mut sys::czk_bases = std::ptr::null_mut();
        let mem = if tables { sys::CZK_MEM_HOST } else { sys::CZK_MEM_HOST | sys::CZK_MEM_NO_TABLES };
        let rc = unsafe { sys::czk_bases_register(ctx.as_ptr(), group, pts.as_ptr(), inf.as_ptr(), inf.len(), mem, &mut h) };
        ctx.expect(rc, "czk_bases_register");
        h
    }

    fn run(group: i32, key: (usize, usize, i32), xy: impl Fn() -> (Vec<u64>, Vec<u8>), scalars: &[Fr], out: &mut [u64]) {
        let ctx = CTX.lock().unwrap();
        let mut map = BASES.lock().unwrap();
        let entry = map.entry(key).or_insert_with(|| {
            let (pts, inf) = xy();
            Registered { handle: register(&ctx, group, &pts, &inf, false), tables: false, uses: 0 }
        });
        entry.uses += 1;
        if entry.uses == 2 && !entry.tables {
            let (pts, inf) = xy();
            unsafe { sys::czk_bases_release(entry.handle) };
            entry.handle = register(&ctx, group, &pts, &inf, true);
            entry.tables = true;
        }
        let s = scalars_to_limbs(scalars);
        let rc = unsafe {
            sys::czk_msm(ctx.as_ptr(), entry.handle, s.as_ptr(), scalars.len(), 1, sys::CZK_SCALAR_MONTGOMERY, sys::CZK_MEM_HOST, out.as_mut_ptr())
        };
        ctx.expect(rc, "czk_msm");
    }

    /// Drop-in body for `<G1Affine as AffineCurve>::multi_scalar_mul`.
    pub fn g1(bases: &[G1Affine], scalars: &[Fr]) -> G1Projective {
        let mut out = [0u64; 18];
        run(sys::CZK_G1, (bases.as_ptr() as usize, bases.len(), sys::CZK_G1), || limbs::g1_bases(bases), scalars, &mut out);
        limbs::g1_from_jac(&out)
    }
    /// Drop-in body for `<G2Affine as AffineCurve>::multi_scalar_mul`.
    pub fn g2(bases: &[G2Affine], scalars: &[Fr]) -> G2Projective {
        let mut out = [0u64; 36];
        run(sys::CZK_G2, (bases.as_ptr() as usize, bases.len(), sys::CZK_G2), || limbs::g2_bases(bases), scalars, &mut out);
        limbs::g2_from_jac(&out)
    }

    /// Several scalar vectors over the same bases in one launch (SPDZ: the `sh` and `mac` MSMs of
    /// `multi_scale_pub_group`, mpc-algebra/src/share/spdz.rs:440-446).  Returns one result per vector.
    pub fn g1_lanes(bases: &[G1Affine], scalar_lanes: &[&[Fr]]) -> Vec<G1Projective> {
        let n = scalar_lanes.iter().map(|l| l.len()).min().unwrap_or(0);
        let lanes = scalar_lanes.len();
        let mut s = vec![0u64; 4 * n * lanes];
        for (ln, v) in scalar_lanes.iter().enumerate() {
            for (i, x) in v.iter().take(n).enumerate() {
                limbs::fr_to(x, &mut s[4 * (ln * n + i)..4 * (ln * n + i) + 4]);
            }
        }
        let mut out = vec![0u64; 18 * lanes];
        let ctx = CTX.lock().unwrap();
        let (pts, inf) = limbs::g1_bases(bases);
        let rc = unsafe {
            sys::czk_msm_g1(ctx.as_ptr(), pts.as_ptr(), inf.as_ptr(), s.as_ptr(), bases.len().min(n), lanes, sys::CZK_SCALAR_MONTGOMERY, out.as_mut_ptr())
        };
        ctx.expect(rc, "czk_msm_g1");
        (0..lanes).map(|ln| limbs::g1_from_jac(&out[18 * ln..18 * ln + 18])).collect()
    }
}

pub mod share {
    //! `MpcField<Fr, SpdzFieldShare<Fr>>` vectors as two Fr lanes (`sh`, `mac`).
    //!
    //! The butterflies of an FFT only add / subtract elements and scale them by PUBLIC twiddles, and those operations act
    //! lane-wise on an SPDZ share (mpc-algebra/src/share/spdz.rs:186-202).  A `Public(x)` entry mixed into a shared vector
    //! behaves like the share `x` would become under `shift` (spdz.rs:204-208, share/add.rs:141-146): the king adds `x` to its
    //! `sh`, every party adds `mac_share() * x` to its `mac` with `mac_share() = 1` on the king and `0` elsewhere
    //! (spdz.rs:30-37).  Lifting every `Public(x)` that way BEFORE the transform gives bit-identical `sh` / `mac` vectors
    //! afterwards.  Values travel as plain `(Fr, Fr)` pairs so that this crate does not depend on
    //! mpc-algebra; the two accessors mpc-algebra gains are in rust/PATCHES.md, patch 3.
    use ark_bls12_377::Fr;
    use ark_ff::Zero;

    /// What `MpcField<Fr, S>` looks like to the shim.
    #[derive(Clone, Copy, Debug, PartialEq, Eq)]
    pub enum Elem {
        Public(Fr),
        /// (sh.val, mac.val) of a `SpdzFieldShare`
        Shared(Fr, Fr),
    }

    /// The two lanes of an element on THIS party.  `am_king` = `Net::am_king()`, `mac_share` = `spdz::mac_share::<Fr>()`.
    #[inline]
    pub fn lift(e: &Elem, am_king: bool, mac_share: &Fr) -> (Fr, Fr) {
        match e {
            Elem::Shared(sh, mac) => (*sh, *mac),
            Elem::Public(x) => (if am_king { *x } else { Fr::zero() }, *mac_share * x),
        }
    }

    /// A vector of elements as the SoA lanes the library transforms: lane 0 = sh, lane 1 = mac, `d` elements each.
    pub struct SpdzLanes {
        pub d: usize,
        pub limbs: Vec<u64>,
    }

    impl SpdzLanes {
        pub fn pack(v: &[Elem], d: usize, am_king: bool, mac_share: &Fr) -> SpdzLanes {
            assert!(v.len() <= d);
            let mut limbs = vec![0u64; 2 * d * 4];
            for (i, e) in v.iter().enumerate() {
                let (sh, mac) = lift(e, am_king, mac_share);
                super::limbs::fr_to(&sh, &mut limbs[4 * i..4 * i + 4]);
                super::limbs::fr_to(&mac, &mut limbs[4 * (d + i)..4 * (d + i) + 4]);
            }
            SpdzLanes { d, limbs }
        }
        /// All-public vectors stay public (the reference's `all_public_or_shared`, wire/field.rs:89-105, keeps them on the
        /// plain path); anything else comes back as shares.
        pub fn unpack(&self) -> Vec<Elem> {
            (0..self.d)
                .map(|i| {
                    Elem::Shared(
                        super::limbs::fr_from(&self.limbs[4 * i..4 * i + 4]),
                        super::limbs::fr_from(&self.limbs[4 * (self.d + i)..4 * (self.d + i) + 4]),
                    )
                })
                .collect()
        }
        /// `{fft, ifft, coset_fft, coset_ifft}_in_place` over both lanes in one call; `in_len` = the vector's length before
        /// `resize(size, zero)`.
        pub fn transform(&mut self, log_size_of_group: u32, kind: super::ntt::Kind, in_len: usize) {
            let ctx = super::CTX.lock().unwrap();
            let k = match kind {
                super::ntt::Kind::Fft => super::sys::CZK_FFT,
                super::ntt::Kind::Ifft => super::sys::CZK_IFFT,
                super::ntt::Kind::CosetFft => super::sys::CZK_COSET_FFT,
                super::ntt::Kind::CosetIfft => super::sys::CZK_COSET_IFFT,
            };
            let rc = unsafe {
                super::sys::czk_ntt_fr(ctx.as_ptr(), self.limbs.as_mut_ptr(), log_size_of_group, 2, k, in_len, super::sys::CZK_MEM_HOST)
            };
            ctx.expect(rc, "czk_ntt_fr (2 share lanes)");
        }
    }
}
