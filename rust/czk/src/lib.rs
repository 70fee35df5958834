//! Safe layer over `czk-sys` for the three seams of alex-ozdemir/collaborative-zksnark.
//!
//! * NTT seam  -- `EvaluationDomain::{fft, ifft, coset_fft, coset_ifft}_in_place`
//!   (algebra/poly/src/domain/radix2/mod.rs:99-117, domain/mod.rs:139-142): [`ntt::transform_in_place`]
//! * MSM seam  -- `AffineCurve::multi_scalar_mul` / `VariableBaseMSM::multi_scalar_mul`
//!   (algebra/ec/src/lib.rs:300-311, algebra/ec/src/msm/variable_base.rs:12-106): [`msm::g1`], [`msm::g2`]
//! * share seam -- `MpcField::{Public, Shared}` vectors as Fr lanes (mpc-algebra/src/wire/field.rs:27-30, 89-105;
//!   share/spdz.rs:31-37, 186-208): [`share::SpdzLanes`]
//! * resident path -- share vectors that stay on the GPU across the witness map and feed the `h` MSM
//!   (mpc-snarks/src/groth/r1cs_to_qap.rs:85-110, mpc-snarks/src/groth/prover.rs:104): [`resident::DeviceLanes`],
//!   [`resident::witness_map`], [`msm::Pinned::msm_resident`] -- `czk_lanes_*` handles, no HIP allocator on the Rust side
//! * opens -- `MpcMultiNet::{broadcast, send_to_king, recv_from_king}` (mpc-net/src/multi.rs:145-242) and
//!   `SpdzFieldShare / AdditiveFieldShare / GszFieldShare::batch_open` (share/spdz.rs:166-185, add.rs:256-259, gsz20/mod.rs:286-300)
//!   between parties that are GPUs: [`net::Net`] -- `czk_net_*`: RCCL over xGMI (or shared memory between processes of one node), on
//!   resident lanes, enqueued on the party's stream
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
    /// czk_ctx_mark: names the work enqueued so far (kernels, `msm_async` calls, deferred downloads).
    pub fn mark(&self) -> u64 {
        let mut m = 0u64;
        let rc = unsafe { sys::czk_ctx_mark(self.raw, &mut m) };
        self.expect(rc, "czk_ctx_mark");
        m
    }
    /// czk_ctx_wait_mark: blocks until the work before `mark` is done and delivers its host results; calls made after the mark keep running
    /// (a transcript point waits for what the transcript absorbs, not for the next round's challenge-independent work).
    pub fn wait_mark(&self, mark: u64) {
        let rc = unsafe { sys::czk_ctx_wait_mark(self.raw, mark) };
        self.expect(rc, "czk_ctx_wait_mark");
    }
    /// czk_ctx_reserve: the twiddle tables of a 2^`ntt_log_d` domain (0: none) and the MSM workspaces for calls of `n_scalars` x `msm_lanes`
    /// on a pinned array (null: none), built at key load instead of inside the first proof (first proof 121 -> 77 ms at 2^20 constraints).
    pub fn reserve(&self, ntt_log_d: u32, ntt_lanes: usize, bases: *const sys::czk_bases, n_scalars: usize, msm_lanes: usize) {
        let rc = unsafe { sys::czk_ctx_reserve(self.raw, ntt_log_d, ntt_lanes, bases, n_scalars, msm_lanes) };
        self.expect(rc, "czk_ctx_reserve");
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

    /// A base slice pinned on the GPU with its window tables: proving-key queries and SRS powers, which outlive many MSMs
    /// (groth16/src/data_structures.rs:132-149, poly-commit/src/kzg10/data_structures.rs).  Created ONCE where the key is loaded
    /// (`Pinned::g1(&pk.a_query[1..])`, rust/PATCHES.md patch 1) and kept next to it; dropping it releases the device copy.
    /// Nothing else is ever cached: a slice that was not pinned runs through the one-shot entry points (no window tables,
    /// bases not kept), so a temporary `Vec` of bases -- `MpcG1Affine::multi_scalar_mul` builds one per call,
    /// mpc-algebra/src/wire/pairing.rs:746-775 -- can neither hit a stale device copy nor leak one.
    pub struct Pinned {
        handle: *mut sys::czk_bases,
        group: i32,
        key: (usize, usize, i32),
        fingerprint: u64,
    }
    unsafe impl Send for Pinned {}

    struct Entry {
        handle: *mut sys::czk_bases,
        fingerprint: u64,
    }
    unsafe impl Send for Entry {}

    lazy_static::lazy_static! {
        // pinned slices by (address, length, group); an entry exists exactly as long as its `Pinned` guard
        static ref PINNED: Mutex<HashMap<(usize, usize, i32), Entry>> = Mutex::new(HashMap::new());
    }

    /// FNV-1a over the limbs of up to 32 evenly spaced points (and the length): a hit on (address, length) is only trusted
    /// when the slice still holds what was pinned -- a guard that outlived its key, or a key mutated in place, falls back to
    /// the one-shot path instead of multiplying with stale bases.
    fn fingerprint(xy: &[u64], words_per_point: usize) -> u64 {
        let n = xy.len() / words_per_point;
        let mut h: u64 = 0xcbf29ce484222325 ^ (n as u64);
        let step = if n > 32 { n / 32 } else { 1 };
        let mut i = 0;
        while i < n {
            for w in &xy[i * words_per_point..(i + 1) * words_per_point] {
                h = (h ^ *w).wrapping_mul(0x100000001b3);
            }
            i += step;
        }
        if n > 0 {
            for w in &xy[(n - 1) * words_per_point..n * words_per_point] {
                h = (h ^ *w).wrapping_mul(0x100000001b3);
            }
        }
        h
    }

    fn sample_g1(bases: &[G1Affine]) -> u64 {
        let n = bases.len();
        let step = if n > 32 { n / 32 } else { 1 };
        let mut idx: Vec<usize> = (0..n).step_by(step).collect();
        if n > 0 {
            idx.push(n - 1);
        }
        let pts: Vec<G1Affine> = idx.iter().map(|&i| bases[i]).collect();
        let (xy, _) = limbs::g1_bases(&pts);
        fingerprint(&xy, 12) ^ (n as u64).rotate_left(17)
    }
    fn sample_g2(bases: &[G2Affine]) -> u64 {
        let n = bases.len();
        let step = if n > 32 { n / 32 } else { 1 };
        let mut idx: Vec<usize> = (0..n).step_by(step).collect();
        if n > 0 {
            idx.push(n - 1);
        }
        let pts: Vec<G2Affine> = idx.iter().map(|&i| bases[i]).collect();
        let (xy, _) = limbs::g2_bases(&pts);
        fingerprint(&xy, 24) ^ (n as u64).rotate_left(17)
    }

    fn scalars_to_limbs(scalars: &[Fr]) -> Vec<u64> {
        let mut s = vec![0u64; 4 * scalars.len()];
        for (i, x) in scalars.iter().enumerate() {
            limbs::fr_to(x, &mut s[4 * i..4 * i + 4]);
        }
        s
    }

    impl Pinned {
        // Lock order everywhere in this module: PINNED before CTX, never the reverse (registration takes them one after the other).
        fn new(group: i32, key: (usize, usize, i32), fp: u64, pts: &[u64], inf: &[u8], flags: i32) -> Pinned {
            let mut h: *mut sys::czk_bases = std::ptr::null_mut();
            {
                let ctx = CTX.lock().unwrap();
                let rc = unsafe { sys::czk_bases_register(ctx.as_ptr(), group, pts.as_ptr(), inf.as_ptr(), inf.len(), sys::CZK_MEM_HOST | flags, &mut h) };
                ctx.expect(rc, "czk_bases_register");
            }
            if let Some(old) = PINNED.lock().unwrap().insert(key, Entry { handle: h, fingerprint: fp }) {
                // the same slice pinned twice: the older guard keeps its handle alive, the map now points at the newer one
                let _ = old;
            }
            Pinned { handle: h, group, key, fingerprint: fp }
        }
        /// Pins a G1 slice (window tables are built once: ~65 ms per 2^20 points).  The slice is taken to hold elements of the
        /// prime-order subgroup -- what `GroupAffine` deserialisation guarantees (short_weierstrass_jacobian.rs:868, :881) and what a proving
        /// key or SRS always is; the bucket kernels then use the curve's twisted Edwards form.  For points of unknown origin use
        /// [`Pinned::g1_checked`] (verifies `[r] P = 0` on the GPU and keeps the complete formulas if any base fails) -- the unpinned
        /// drop-in path ([`g1`]) makes no such assumption at all.
        pub fn g1(bases: &[G1Affine]) -> Pinned {
            let (pts, inf) = limbs::g1_bases(bases);
            Pinned::new(sys::CZK_G1, (bases.as_ptr() as usize, bases.len(), sys::CZK_G1), sample_g1(bases), &pts, &inf, 0)
        }
        /// As [`Pinned::g1`], with `is_in_correct_subgroup_assuming_on_curve` (short_weierstrass_jacobian.rs:131) run over the slice at
        /// registration (CZK_MEM_CHECK_SUBGROUP); `bad_bases()` reports how many failed.
        pub fn g1_checked(bases: &[G1Affine]) -> Pinned {
            let (pts, inf) = limbs::g1_bases(bases);
            Pinned::new(sys::CZK_G1, (bases.as_ptr() as usize, bases.len(), sys::CZK_G1), sample_g1(bases), &pts, &inf, sys::CZK_MEM_CHECK_SUBGROUP)
        }
        pub fn g2(bases: &[G2Affine]) -> Pinned {
            let (pts, inf) = limbs::g2_bases(bases);
            Pinned::new(sys::CZK_G2, (bases.as_ptr() as usize, bases.len(), sys::CZK_G2), sample_g2(bases), &pts, &inf, 0)
        }
        /// Number of pinned bases outside the prime-order subgroup (czk_bases_check_subgroup; runs the check now unless registration did).
        pub fn bad_bases(&self) -> usize {
            let mut bad: usize = 0;
            let ctx = CTX.lock().unwrap();
            let rc = unsafe { sys::czk_bases_check_subgroup(ctx.as_ptr(), self.handle, &mut bad) };
            ctx.expect(rc, "czk_bases_check_subgroup");
            bad
        }
        pub fn as_ptr(&self) -> *mut sys::czk_bases {
            self.handle
        }
        /// The MSM over share lanes that are already on the GPU (`h` out of the witness map, prover.rs:104), enqueue-only.  The library
        /// keeps the destination pointer until the next synchronisation, so the destination is OWNED by the returned [`PendingMsm`]: a
        /// heap buffer that cannot be dropped or moved before the results are delivered (dropping the guard synchronises first; leaking it
        /// with `mem::forget` leaks the buffer, which stays valid).  `wait()` yields one Jacobian result per lane, 18 (G1) or 36 (G2) limbs.
        /// Borrowing `self` keeps the pinned handle alive for as long as the MSM is in flight.
        pub fn msm_resident<'a>(&'a self, scalars: &super::resident::DeviceLanes, n_scalars: usize) -> PendingMsm<'a> {
            let jw = if self.group == sys::CZK_G1 { 18 } else { 36 };
            let mut buf = vec![0u64; jw * scalars.lanes()].into_boxed_slice();
            let ctx = CTX.lock().unwrap();
            let rc = unsafe {
                sys::czk_msm_async(ctx.as_ptr(), self.handle, scalars.data(0, 0), n_scalars, scalars.lanes(), sys::CZK_SCALAR_MONTGOMERY,
                                   sys::CZK_MEM_DEVICE, buf.as_mut_ptr())
            };
            ctx.expect(rc, "czk_msm_async");
            PendingMsm { buf: Some(buf), _bases: std::marker::PhantomData }
        }
    }

    /// Results of an enqueued MSM (see [`Pinned::msm_resident`]).
    pub struct PendingMsm<'a> {
        buf: Option<Box<[u64]>>,
        _bases: std::marker::PhantomData<&'a Pinned>,
    }
    impl<'a> PendingMsm<'a> {
        /// Synchronises the context (delivers every pending MSM of it) and returns the Jacobian limbs.
        pub fn wait(mut self) -> Box<[u64]> {
            CTX.lock().unwrap().sync();
            self.buf.take().unwrap()
        }
    }
    impl<'a> Drop for PendingMsm<'a> {
        fn drop(&mut self) {
            if self.buf.is_some() {
                CTX.lock().unwrap().sync();   // the library may still hold the pointer: deliver before the buffer is freed
            }
        }
    }

    impl Drop for Pinned {
        fn drop(&mut self) {
            // released UNDER the PINNED lock: `g1` / `g2` hold that lock for the whole czk_msm call on a looked-up handle, so a guard dropped on
            // another thread cannot free a handle that a call is using
            let mut map = PINNED.lock().unwrap();
            if let Some(e) = map.get(&self.key) {
                if e.handle == self.handle {
                    map.remove(&self.key);
                }
            }
            let _ = self.fingerprint;
            unsafe { sys::czk_bases_release(self.handle) };
        }
    }

    type PinnedMap<'a> = std::sync::MutexGuard<'a, HashMap<(usize, usize, i32), Entry>>;
    /// The pinned handle of `key` if the slice still holds what was pinned -- together with the PINNED lock, which the caller keeps until
    /// its call on the handle has returned.
    fn lookup<'a>(key: (usize, usize, i32), fp: impl Fn() -> u64) -> (PinnedMap<'a>, Option<*mut sys::czk_bases>) {
        let map = PINNED.lock().unwrap();
        let h = match map.get(&key) {
            Some(e) if e.fingerprint == fp() => Some(e.handle),
            _ => None,
        };
        (map, h)
    }

    /// Drop-in body for `<G1Affine as AffineCurve>::multi_scalar_mul`: pinned slices use their tables, everything else the
    /// one-shot form (`czk_msm_g1`: bases copied, no tables, nothing kept, and -- like the reference's own function -- correct for ANY
    /// curve points: the one-shot entry points register with CZK_MEM_ANY_POINTS).
    pub fn g1(bases: &[G1Affine], scalars: &[Fr]) -> G1Projective {
        let mut out = [0u64; 18];
        let s = scalars_to_limbs(scalars);
        let (_pinned_lock, pinned) = lookup((bases.as_ptr() as usize, bases.len(), sys::CZK_G1), || sample_g1(bases));
        let ctx = CTX.lock().unwrap();
        let rc = match pinned {
            Some(h) => unsafe {
                sys::czk_msm(ctx.as_ptr(), h, s.as_ptr(), scalars.len(), 1, sys::CZK_SCALAR_MONTGOMERY, sys::CZK_MEM_HOST, out.as_mut_ptr())
            },
            None => {
                let (pts, inf) = limbs::g1_bases(bases);
                let n = bases.len().min(scalars.len());
                unsafe { sys::czk_msm_g1(ctx.as_ptr(), pts.as_ptr(), inf.as_ptr(), s.as_ptr(), n, 1, sys::CZK_SCALAR_MONTGOMERY, out.as_mut_ptr()) }
            }
        };
        ctx.expect(rc, "czk_msm (G1)");
        limbs::g1_from_jac(&out)
    }
    /// Drop-in body for `<G2Affine as AffineCurve>::multi_scalar_mul`.
    pub fn g2(bases: &[G2Affine], scalars: &[Fr]) -> G2Projective {
        let mut out = [0u64; 36];
        let s = scalars_to_limbs(scalars);
        let (_pinned_lock, pinned) = lookup((bases.as_ptr() as usize, bases.len(), sys::CZK_G2), || sample_g2(bases));
        let ctx = CTX.lock().unwrap();
        let rc = match pinned {
            Some(h) => unsafe {
                sys::czk_msm(ctx.as_ptr(), h, s.as_ptr(), scalars.len(), 1, sys::CZK_SCALAR_MONTGOMERY, sys::CZK_MEM_HOST, out.as_mut_ptr())
            },
            None => {
                let (pts, inf) = limbs::g2_bases(bases);
                let n = bases.len().min(scalars.len());
                unsafe { sys::czk_msm_g2(ctx.as_ptr(), pts.as_ptr(), inf.as_ptr(), s.as_ptr(), n, 1, sys::CZK_SCALAR_MONTGOMERY, out.as_mut_ptr()) }
            }
        };
        ctx.expect(rc, "czk_msm (G2)");
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

pub mod resident {
    //! Share vectors that stay on the GPU: what `Vec<T>` is to the reference's `witness_map`, which keeps `a`, `b`, `c` alive
    //! across seven transforms and hands `h` to the MSM (mpc-snarks/src/groth/r1cs_to_qap.rs:85-110, groth/prover.rs:104).
    //! A `DeviceLanes` is a `czk_lanes` handle: `lanes` x `capacity` Fr in HBM, allocated and freed by the library -- the Rust
    //! side needs no HIP allocator.  Compiled and GPU-tested as C++: `czk::DeviceLanes` / `czk::R1CStoQAP::witness_map`
    //! (include/czk.hpp), driven by tools/host_demo.cpp `bench` at the full benchmark size.
    use super::{sys, CTX};
    use ark_bls12_377::Fr;

    pub struct DeviceLanes {
        raw: *mut sys::czk_lanes,
        /// the logical `Vec::len()` of every lane; elements beyond it are the zero padding `resize` would add
        pub len: usize,
    }
    unsafe impl Send for DeviceLanes {}

    impl DeviceLanes {
        /// `vec![T::zero(); capacity]` for `lanes` share components (r1cs_to_qap.rs:66-67)
        pub fn zeros(lanes: usize, capacity: usize) -> DeviceLanes {
            let ctx = CTX.lock().unwrap();
            let mut raw: *mut sys::czk_lanes = std::ptr::null_mut();
            let rc = unsafe { sys::czk_lanes_alloc(ctx.as_ptr(), lanes, capacity, &mut raw) };
            ctx.expect(rc, "czk_lanes_alloc");
            DeviceLanes { raw, len: capacity }
        }
        pub fn lanes(&self) -> usize {
            unsafe { sys::czk_lanes_count(self.raw) }
        }
        pub fn capacity(&self) -> usize {
            unsafe { sys::czk_lanes_len(self.raw) }
        }
        /// Device address of one element, for the `CZK_MEM_DEVICE` arguments of `czk_sys`.
        pub fn data(&self, lane: usize, elem: usize) -> *mut u64 {
            unsafe { sys::czk_lanes_data(self.raw, lane, elem) }
        }
        /// Host values into lane `lane` from element `elem` on; returns when `v` has been read (it may be dropped).
        pub fn upload(&mut self, lane: usize, elem: usize, v: &[Fr]) {
            let mut l = vec![0u64; 4 * v.len()];
            for (i, x) in v.iter().enumerate() {
                super::limbs::fr_to(x, &mut l[4 * i..4 * i + 4]);
            }
            let ctx = CTX.lock().unwrap();
            let rc = unsafe { sys::czk_lanes_upload(ctx.as_ptr(), self.raw, lane, elem, l.as_ptr(), v.len()) };
            ctx.expect(rc, "czk_lanes_upload");
        }
        pub fn download(&self, lane: usize, elem: usize, n: usize) -> Vec<Fr> {
            let mut l = vec![0u64; 4 * n];
            let ctx = CTX.lock().unwrap();
            let rc = unsafe { sys::czk_lanes_download(ctx.as_ptr(), self.raw, lane, elem, l.as_mut_ptr(), n) };
            ctx.expect(rc, "czk_lanes_download");
            (0..n).map(|i| super::limbs::fr_from(&l[4 * i..4 * i + 4])).collect()
        }
        /// `{fft, ifft, coset_fft, coset_ifft}_in_place` on every lane, in place in HBM.
        pub fn transform(&mut self, log_size_of_group: u32, kind: super::ntt::Kind) {
            assert!(self.len <= 1usize << log_size_of_group && self.capacity() == 1usize << log_size_of_group);
            let k = match kind {
                super::ntt::Kind::Fft => sys::CZK_FFT,
                super::ntt::Kind::Ifft => sys::CZK_IFFT,
                super::ntt::Kind::CosetFft => sys::CZK_COSET_FFT,
                super::ntt::Kind::CosetIfft => sys::CZK_COSET_IFFT,
            };
            let ctx = CTX.lock().unwrap();
            let rc = unsafe { sys::czk_ntt_fr(ctx.as_ptr(), self.data(0, 0), log_size_of_group, self.lanes(), k, self.len, sys::CZK_MEM_DEVICE) };
            ctx.expect(rc, "czk_ntt_fr (resident lanes)");
            self.len = self.capacity();
        }
    }

    impl Drop for DeviceLanes {
        fn drop(&mut self) {
            unsafe { sys::czk_lanes_free(self.raw) }
        }
    }

    /// `R1CStoQAP::witness_map` from the first `ifft` to `h` on resident lanes (r1cs_to_qap.rs:85-110).  `a`, `b`, `c`: the
    /// evaluated constraint rows (`len` = rows, capacity = domain size).  `batch_product(a, b, ab)` is
    /// `F::batch_product_in_place` (:92): for shares the Beaver protocol -- its opens travel over mpc-net from
    /// `DeviceLanes::download` / `upload` of the `sh` lane, its local half is `czk_fr_beaver_combine`; for a single prover
    /// `czk_fr_vec_op(CZK_OP_MUL)`.  `ab` = h on return.
    pub fn witness_map<P: FnOnce(&mut DeviceLanes, &mut DeviceLanes, &mut DeviceLanes)>(
        log_size_of_group: u32,
        a: &mut DeviceLanes,
        b: &mut DeviceLanes,
        c: &mut DeviceLanes,
        ab: &mut DeviceLanes,
        batch_product: P,
    ) {
        let d = 1usize << log_size_of_group;
        let lanes = a.lanes();
        assert!(a.capacity() == d && b.capacity() == d && c.capacity() == d && ab.capacity() == d);
        assert!(b.lanes() == lanes && c.lanes() == lanes && ab.lanes() == lanes);
        {
            let ctx = CTX.lock().unwrap();
            let rc = unsafe { sys::czk_witness_map_pre(ctx.as_ptr(), a.data(0, 0), a.len, b.data(0, 0), b.len, log_size_of_group, lanes) };
            ctx.expect(rc, "czk_witness_map_pre");
        }
        a.len = d;
        b.len = d;
        batch_product(a, b, ab);
        {
            let ctx = CTX.lock().unwrap();
            let rc = unsafe { sys::czk_witness_map_post(ctx.as_ptr(), ab.data(0, 0), c.data(0, 0), c.len, log_size_of_group, lanes) };
            ctx.expect(rc, "czk_witness_map_post");
        }
        ab.len = d;
        c.len = d;
    }
}

pub mod net {
    //! mpc-net between GPUs.  The reference's parties exchange share vectors through the process-global `MpcMultiNet`
    //! (mpc-net/src/multi.rs:15-23, one TCP connection per pair); with the share lanes resident in HBM the same four primitives and
    //! the three batch opens run through a `czk_net` communicator instead -- RCCL over xGMI when every party owns a GPU, shared
    //! memory when parties share one -- enqueued on the party's stream, so `witness_map` never leaves HBM.  The
    //! communicator id travels over the reference's own sockets once, at start-up (`Net::init_rccl`).
    //! Compiled and GPU-tested as C++: `czk::Net` (include/czk.hpp), `tools/host_demo.cpp party`.
    use super::resident::DeviceLanes;
    use super::{sys, CTX};
    use ark_bls12_377::Fr;
    use std::ffi::CStr;
    use std::os::raw::c_int;

    pub struct Net {
        raw: *mut sys::czk_net,
    }
    // A `Net` may move to another thread: every method that reaches the shared context (its stream, its staging buffers, its error string) takes the
    // `CTX` mutex first, exactly like the transforms and MSMs of this crate; `party_id` / `n_parties` / `stats` only read the communicator's own fields.
    unsafe impl Send for Net {}

    /// mpc-net/src/lib.rs `Stats`
    #[derive(Debug, Default, Clone, Copy)]
    pub struct Stats {
        pub bytes_sent: u64,
        pub bytes_recv: u64,
        pub broadcasts: u64,
        pub to_king: u64,
        pub from_king: u64,
    }

    impl Net {
        /// rank 0 only: the 128 bytes the other parties need for `init_rccl` -- send them with `MpcMultiNet::recv_from_king`
        pub fn unique_id_rccl() -> Vec<u8> {
            let mut id = vec![0u8; sys::CZK_NET_UNIQUE_ID_BYTES as usize];
            let mut len = 0usize;
            let rc = unsafe { sys::czk_net_unique_id(sys::CZK_NET_RCCL, id.as_mut_ptr(), id.len(), &mut len) };
            assert!(rc == sys::CZK_OK, "czk_net_unique_id: status {} (librccl.so.1 not loadable?)", rc);
            id.truncate(len);
            id
        }
        /// one party per GPU: `party_id` / `n_parties` as `MpcMultiNet` reports them, `id` from rank 0's `unique_id_rccl`
        pub fn init_rccl(party_id: usize, n_parties: usize, id: &[u8]) -> Net {
            Net::create(sys::CZK_NET_RCCL, party_id, n_parties, id)
        }
        /// parties are processes of one node in any assignment to GPUs (tests, rigs with fewer GPUs than parties): `id` = 1..32 bytes
        pub fn init_shm(party_id: usize, n_parties: usize, id: &[u8]) -> Net {
            Net::create(sys::CZK_NET_SHM, party_id, n_parties, id)
        }
        /// the same with device mailboxes mapped between the processes (hipIpc): an exchange never leaves device memory
        pub fn init_ipc(party_id: usize, n_parties: usize, id: &[u8]) -> Net {
            Net::create(sys::CZK_NET_IPC, party_id, n_parties, id)
        }
        fn create(transport: c_int, party_id: usize, n_parties: usize, id: &[u8]) -> Net {
            // The rendezvous blocks until every party has arrived (up to the communicator's timeout): the context lock is NOT held across it -- other
            // threads keep proving.  czk_net_create itself only reads the context's device index (and writes its error string if it fails).
            let ctx_ptr = { CTX.lock().unwrap().as_ptr() };
            let mut raw: *mut sys::czk_net = std::ptr::null_mut();
            let rc = unsafe { sys::czk_net_create(ctx_ptr, transport, party_id as c_int, n_parties as c_int, id.as_ptr(), id.len(), &mut raw) };
            CTX.lock().unwrap().expect(rc, "czk_net_create");
            Net { raw }
        }
        fn expect(&self, rc: c_int, what: &str) {
            if rc != sys::CZK_OK {
                let msg = unsafe { CStr::from_ptr(sys::czk_net_last_error(self.raw)) }.to_string_lossy().into_owned();
                panic!("{}: czk status {}: {}", what, rc, msg);
            }
        }
        pub fn party_id(&self) -> usize {
            unsafe { sys::czk_net_rank(self.raw) as usize }
        }
        pub fn n_parties(&self) -> usize {
            unsafe { sys::czk_net_world(self.raw) as usize }
        }
        pub fn am_king(&self) -> bool {
            self.party_id() == 0
        }
        pub fn stats(&self) -> Stats {
            let mut s = [0u64; 5];
            let rc = unsafe { sys::czk_net_stats(self.raw, s.as_mut_ptr()) };
            self.expect(rc, "czk_net_stats");
            Stats { bytes_sent: s[0], bytes_recv: s[1], broadcasts: s[2], to_king: s[3], from_king: s[4] }
        }
        /// `Net::broadcast_bytes` on host bytes (mpc-net/src/lib.rs:44-47)
        pub fn broadcast_bytes(&self, out: &[u8]) -> Vec<Vec<u8>> {
            let mut flat = vec![0u8; out.len() * self.n_parties()];
            let _ctx = CTX.lock().unwrap();   // the communicator enqueues on the shared context's stream and uses its staging buffers
            let rc = unsafe { sys::czk_net_broadcast(self.raw, out.as_ptr() as *const _, out.len(), flat.as_mut_ptr() as *mut _, sys::CZK_MEM_HOST) };
            self.expect(rc, "czk_net_broadcast");
            flat.chunks(out.len().max(1)).take(self.n_parties()).map(|c| c.to_vec()).collect()
        }
        /// `SpdzFieldShare::batch_open` (share/spdz.rs:166-185) on resident lanes: lane 0 of `shares` = sh, lane 1 = mac; the opened
        /// (public) vector lands in lane 0 of `out`.  Panics where the reference asserts (MAC check).
        pub fn spdz_batch_open(&self, shares: &DeviceLanes, mac_share: &Fr, n: usize, out: &mut DeviceLanes, commit: bool) {
            let mut ms = [0u64; 4];
            super::limbs::fr_to(mac_share, &mut ms);
            let mut bad = 0u64;
            let _ctx = CTX.lock().unwrap();   // the communicator enqueues on the shared context's stream and uses its staging buffers
            let rc = unsafe {
                sys::czk_spdz_batch_open(self.raw, shares.data(0, 0), shares.data(1, 0), ms.as_ptr(), n, out.data(0, 0),
                                         if commit { sys::CZK_OPEN_COMMIT } else { 0 }, &mut bad)
            };
            self.expect(rc, "czk_spdz_batch_open");
            assert!(bad == 0, "assertion failed: sum.is_zero() (SPDZ MAC check on {} values)", bad);
        }
        /// `AdditiveFieldShare::batch_open` (share/add.rs:256-259)
        pub fn add_batch_open(&self, val: &DeviceLanes, n: usize, out: &mut DeviceLanes) {
            let _ctx = CTX.lock().unwrap();   // the communicator enqueues on the shared context's stream and uses its staging buffers
            let rc = unsafe { sys::czk_add_batch_open(self.raw, val.data(0, 0), n, out.data(0, 0)) };
            self.expect(rc, "czk_add_batch_open");
        }
        /// `GszFieldShare::batch_open` (share/gsz20/mod.rs:286-300) with one degree bound
        pub fn gsz_batch_open(&self, val: &DeviceLanes, n: usize, degree: u32, out: &mut DeviceLanes) {
            let mut bad = 0u64;
            let _ctx = CTX.lock().unwrap();   // the communicator enqueues on the shared context's stream and uses its staging buffers
            let rc = unsafe { sys::czk_gsz_batch_open(self.raw, val.data(0, 0), n, std::ptr::null(), degree, out.data(0, 0), &mut bad) };
            self.expect(rc, "czk_gsz_batch_open");
            assert!(bad == 0, "assertion failed: p.degree() <= d ({} values)", bad);
        }
        /// `gsz20::batch_king_compute(shares, new_degree, |r| r)` (share/gsz20/mod.rs:494-527): the degree reduction inside `batch_mult`
        pub fn gsz_batch_king_compute(&self, val: &DeviceLanes, n: usize, degree: u32, out: &mut DeviceLanes) {
            let mut bad = 0u64;
            let _ctx = CTX.lock().unwrap();   // the communicator enqueues on the shared context's stream and uses its staging buffers
            let rc = unsafe { sys::czk_gsz_batch_king_compute(self.raw, val.data(0, 0), n, std::ptr::null(), degree, out.data(0, 0), &mut bad) };
            self.expect(rc, "czk_gsz_batch_king_compute");
            assert!(bad == 0, "assertion failed: p.degree() <= d on the king ({} values)", bad);
        }
    }

    impl Drop for Net {
        fn drop(&mut self) {
            unsafe { sys::czk_net_destroy(self.raw) }
        }
    }
}
