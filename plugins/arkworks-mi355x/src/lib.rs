//! Groth16 [`ProofSystem`] for OpenZL with the prover's hot path (5 MSMs + 7 NTTs behind `ark_groth16::create_random_proof`,
//! reached from `plugins/arkworks/src/groth16.rs:454`) on an AMD MI355X through `libzl_backend.so`.
//!
//! **SOURCE ONLY -- never compiled or tested**: the build image of this repository has no `cargo` / `rustc`.  What *is*
//! tested is the C ABI this file binds (`include/zl_backend.h`, driven through ctypes by `tests/test_gpu_*.py`), and `src/ffi.rs` is
//! generated from that header (`tools/gen_rust_ffi.py`; `tests/test_abi.py::test_rust_ffi_matches_header` keeps them in step).
//!
//! Same trait surface as the reference's `Groth16<E>` (`plugins/arkworks/src/groth16.rs:405-467`):
//! `Compiler = R1CS<E::Fr>`, `PublicParameters = ()`, `Input = Vec<E::Fr>`, `Proof = Proof<E>`, `VerifyingContext<E>`, opaque `Error`,
//! so the ECLAIR / gadget stack above it is unchanged.  Only `ProvingContext` differs: it also owns the device-resident copy of the key.
//!
//! * `compile`: arkworks' own circuit-specific setup on the CPU (one-time), then the five query vectors of the proving key go to the device AS THEY LIE IN
//!   MEMORY: `zl_bases_upload(.., size_of::<GroupAffine>, offset of `infinity`, ZL_MONT, ..)` -- no dependence on a byte layout recalled from
//!   `serialize_unchecked` (round 6; the same call with a Rust-like record layout is driven from C by `tests/c/inmemory_key.c`, G1 and G2).
//!   With the cargo feature `wire-upload` the key travels as `ProvingKey::serialize_unchecked` bytes through `zl_groth16_keys_from_bytes` instead.
//! * threads: a `ProvingContext` owns its root `zl_ctx` and a pool of prover lanes (`zl_ctx_fork`); it is `Send + Sync` like the reference's
//!   (groth16.rs:127-140): N threads proving through one `&ProvingContext` run on N lanes over ONE device-resident key.
//! * `prove`: the constraint matrices are uploaded once per context (`zl_r1cs_upload`), every proof ships only the assignment in
//!   arkworks' in-memory Montgomery limbs (`zl_groth16_prove_resident`, `ZL_MONT`) and the two blinding scalars sampled here with
//!   `E::Fr::rand`, exactly as `create_random_proof` samples them.
//! * `verify`: arkworks' own (`verify_with_processed_vk`), unchanged.
#![allow(clippy::missing_safety_doc)]

pub mod ffi;

use ark_ec::{AffineCurve, PairingEngine};
use ark_ff::{BigInteger, Field, FromBytes, PrimeField, UniformRand, Zero};
use ark_groth16::{Groth16 as ArkGroth16, Proof as ArkProof, ProvingKey};
use ark_relations::r1cs::{ConstraintSynthesizer, ConstraintSystem, OptimizationGoal, SynthesisMode};
use ark_serialize::{CanonicalDeserialize, CanonicalSerialize, Read, SerializationError, Write};
use ark_snark::SNARK;
use core::{cell::Cell, marker::PhantomData, ptr, sync::atomic::{AtomicBool, Ordering}};
use std::sync::Mutex;
use openzl_crypto::constraint::ProofSystem;
use openzl_plugin_arkworks::{
    constraint::R1CS,
    groth16::{Error, Proof, VerifyingContext},
    serialize::{ArkReader, ArkWriter},
};
use openzl_util::{
    codec::{self, DecodeError},
    rand::{CryptoRng, RngCore, SizedRng},
};

/// The two pairing engines the backend is built for (include/zl_backend.h: `zl_curve_t`).
///
/// `PairingEngine` has no generic way to build an affine point from coordinates (ark-ec 0.3.0: `E::G1Affine` is only bound by `AffineCurve`), so the
/// constructors live here, written against the concrete curve crates: `short_weierstrass_jacobian::GroupAffine::new(x, y, infinity)` and, for the twist,
/// `QuadExtField::new(c0, c1)` coordinates (ark-ec 0.3.0 `models/short_weierstrass_jacobian.rs`, ark-ff 0.3.0 `fields/models/quadratic_extension.rs`).
pub trait Mi355xEngine: PairingEngine {
    /// `ZL_BLS12_381` or `ZL_BN254`
    const CURVE: i32;
    /// u64 limbs per base-field element in the ABI layouts (6 / 4)
    const FQ_LIMBS: usize;
    /// the finite G1 point (x, y)
    fn g1_from_xy(x: Self::Fq, y: Self::Fq) -> Self::G1Affine;
    /// the finite G2 point (x0 + x1 u, y0 + y1 u)
    fn g2_from_xy(x0: Self::Fq, x1: Self::Fq, y0: Self::Fq, y1: Self::Fq) -> Self::G2Affine;
    /// byte offset of `infinity: bool` inside one in-memory `G1Affine` / `G2Affine` record, MEASURED on a value (no layout is assumed: `GroupAffine` has no `#[repr]`)
    fn g1_infinity_offset() -> usize;
    fn g2_infinity_offset() -> usize;
    /// canonical limbs x || y (G1) and x.c0 || x.c1 || y.c0 || y.c1 (G2) of a finite point, all-zero for infinity (zl_g16_pk's single points)
    fn g1_canonical(p: &Self::G1Affine) -> Vec<u64>;
    fn g2_canonical(p: &Self::G2Affine) -> Vec<u64>;
}
macro_rules! engine_memory_layout {
    ($g1:ty, $g2:ty) => {
        fn g1_infinity_offset() -> usize {
            let p = <$g1>::zero();
            (&p.infinity as *const bool as usize) - (&p as *const $g1 as usize)
        }
        fn g2_infinity_offset() -> usize {
            let p = <$g2>::zero();
            (&p.infinity as *const bool as usize) - (&p as *const $g2 as usize)
        }
        fn g1_canonical(p: &Self::G1Affine) -> Vec<u64> {
            let mut v = vec![0u64; 2 * Self::FQ_LIMBS];
            if !p.infinity {
                v[..Self::FQ_LIMBS].copy_from_slice(p.x.into_repr().as_ref());
                v[Self::FQ_LIMBS..].copy_from_slice(p.y.into_repr().as_ref());
            }
            v
        }
        fn g2_canonical(p: &Self::G2Affine) -> Vec<u64> {
            let n = Self::FQ_LIMBS;
            let mut v = vec![0u64; 4 * n];
            if !p.infinity {
                for (k, c) in [p.x.c0, p.x.c1, p.y.c0, p.y.c1].iter().enumerate() {
                    v[k * n..(k + 1) * n].copy_from_slice(c.into_repr().as_ref());
                }
            }
            v
        }
    };
}
impl Mi355xEngine for ark_bls12_381::Bls12_381 {
    const CURVE: i32 = ffi::ZL_BLS12_381;
    const FQ_LIMBS: usize = 6;
    #[inline]
    fn g1_from_xy(x: Self::Fq, y: Self::Fq) -> Self::G1Affine {
        ark_bls12_381::G1Affine::new(x, y, false)
    }
    #[inline]
    fn g2_from_xy(x0: Self::Fq, x1: Self::Fq, y0: Self::Fq, y1: Self::Fq) -> Self::G2Affine {
        ark_bls12_381::G2Affine::new(ark_bls12_381::Fq2::new(x0, x1), ark_bls12_381::Fq2::new(y0, y1), false)
    }
    engine_memory_layout!(ark_bls12_381::G1Affine, ark_bls12_381::G2Affine);
}
impl Mi355xEngine for ark_bn254::Bn254 {
    const CURVE: i32 = ffi::ZL_BN254;
    const FQ_LIMBS: usize = 4;
    #[inline]
    fn g1_from_xy(x: Self::Fq, y: Self::Fq) -> Self::G1Affine {
        ark_bn254::G1Affine::new(x, y, false)
    }
    #[inline]
    fn g2_from_xy(x0: Self::Fq, x1: Self::Fq, y0: Self::Fq, y1: Self::Fq) -> Self::G2Affine {
        ark_bn254::G2Affine::new(ark_bn254::Fq2::new(x0, x1), ark_bn254::Fq2::new(y0, y1), false)
    }
    engine_memory_layout!(ark_bn254::G1Affine, ark_bn254::G2Affine);
}

std::thread_local! {
    /// One `zl_ctx` per thread for the FREE functions below (`msm_g1`, `upload_g1_bases`, `ntt_in_place`): a ctx is bound to one GPU / stream and used from
    /// one thread at a time (zl_backend.h, "Conventions").  A `ProvingContext` does not use it: it owns its root ctx and its lanes.
    static CTX: Cell<*mut ffi::zl_ctx> = Cell::new(ptr::null_mut());
}

/// The calling thread's context on device `ZL_DEVICE` (default 0), created on first use.
fn ctx() -> Result<*mut ffi::zl_ctx, Error> {
    CTX.with(|c| {
        if c.get().is_null() {
            let device = std::env::var("ZL_DEVICE").ok().and_then(|v| v.parse().ok()).unwrap_or(0);
            let mut p = ptr::null_mut();
            // every failure collapses into the plugin's opaque Error, like `.map_err(|_| Error)` at groth16.rs:438-465
            if unsafe { ffi::zl_ctx_create(&mut p, device) } != ffi::ZL_OK {
                return Err(Error);
            }
            c.set(p);
        }
        Ok(c.get())
    })
}

/// Proving context: the arkworks proving key (what the reference's `ProvingContext<E>(pub ProvingKey<E>)` is, groth16.rs:120-140) and its
/// device-resident twin.  Like the reference's it is `Clone + Debug + Eq`, `CanonicalSerialize / CanonicalDeserialize`,
/// `codec::Encode / Decode` (all of them delegate to `key`; the device state is rebuilt from it: clone / decode = one upload) -- and, since round 6,
/// `Send + Sync`: the context OWNS its root `zl_ctx` (the five queries and the constraint matrices live there) and a pool of prover lanes
/// (`zl_ctx_fork`: a lane has its own streams, scratch and host workers and READS the root's device-resident objects).  `prove` takes an idle lane --
/// or forks one more -- for the duration of one proof, so N threads sharing `&ProvingContext` prove side by side over ONE copy of the key
/// (tests/test_gpu_lanes.py exercises exactly this through the C ABI: three threads x 8 proofs, byte-identical; two lanes: +6 % proofs/s at
/// 958 465 constraints, +64 % at 235).  The root ctx itself never proves: it only takes uploads, under the state mutex (a `zl_ctx` is single-caller).
///
/// A proving key belongs to ONE circuit (in the reference too: a key used with another circuit yields a proof that does not verify), so the
/// constraint matrices are uploaded with the first proof and stay on the device.  `shape` records that circuit's (constraints, instance
/// variables, witness variables, linear combinations): a later compiler with another shape is refused with `Error` instead of being proved
/// against the resident matrices.  What the shape check cannot see -- a DIFFERENT circuit with the SAME four counts -- is covered three ways:
/// * `digest` holds an FNV-1a fingerprint of all rows of A, B, C taken at upload; with [`ProvingContext::set_check_binding`]`(true)` -- the default of debug
///   builds -- every proof rebuilds the matrices of its compiler, fingerprints them and refuses (`Error`) on a mismatch;
/// * [`ProvingContext::rebind`] drops the resident matrices, so that the next proof uploads (and fingerprints) its own;
/// * and the contract, as in the reference: `prove` NEVER checks the proof it returns (`Groth16::prove`, groth16.rs:445-457, does not either).
pub struct ProvingContext<E>
where
    E: Mi355xEngine,
{
    /// The key as arkworks holds it
    pub key: ProvingKey<E>,
    /// the ctx the key's device objects live on (owned)
    root: *mut ffi::zl_ctx,
    /// bases handles of a_query, b_g1_query, h_query, l_query (G1) and b_g2_query (G2) on `root`
    queries: [u64; 5],
    /// canonical limbs of alpha_g1, beta_g1, delta_g1 (x || y) and beta_g2, delta_g2 (x.c0 || x.c1 || y.c0 || y.c1): what `zl_g16_pk` points at
    points: [Vec<u64>; 5],
    /// only with the `wire-upload` feature: the key object `zl_groth16_keys_from_bytes` built (owns the handles in `queries`); null otherwise
    keys: *mut ffi::zl_g16_keys,
    state: Mutex<State>,
    /// fingerprint the compiler of every proof against `digest` (costs one `to_matrices` per proof)
    check_binding: AtomicBool,
}
struct State {
    /// idle prover lanes (forks of `root`)
    lanes: Vec<*mut ffi::zl_ctx>,
    /// `zl_r1cs_upload` handle of the circuit this key was compiled for (0 until the first proof), its shape and the FNV-1a fingerprint of its rows
    r1cs: u64,
    shape: [usize; 4],
    digest: u64,
}
// Safety: `root` and the lanes are only ever used by one thread at a time -- `root` under `state`'s mutex, a lane by the thread that popped it from the
// pool -- which is the C ABI's rule for a zl_ctx (include/zl_backend.h: "single-caller"); the device objects behind `queries` / `r1cs` are read-only while
// lanes exist (the backend refuses to free or re-table them: zl_ctx_fork); `key` and `points` are immutable after `new`.
unsafe impl<E> Send for ProvingContext<E> where E: Mi355xEngine {}
unsafe impl<E> Sync for ProvingContext<E> where E: Mi355xEngine {}

/// a lane borrowed from the pool for one proof; goes back on drop (also on the error paths)
struct LaneGuard<'a, E: Mi355xEngine> {
    context: &'a ProvingContext<E>,
    lane: *mut ffi::zl_ctx,
}
impl<'a, E: Mi355xEngine> Drop for LaneGuard<'a, E> {
    fn drop(&mut self) {
        if let Ok(mut st) = self.context.state.lock() {
            st.lanes.push(self.lane);
        }
    }
}

/// FNV-1a over the CSR arrays of the three matrices, in upload order: row pointers, columns, coefficient limbs
fn csr_digest(parts: [(&[u32], &[u32], &[u64]); 3]) -> u64 {
    let mut h = 0xcbf2_9ce4_8422_2325u64;
    let mut eat = |v: u64| {
        for b in v.to_le_bytes() {
            h ^= b as u64;
            h = h.wrapping_mul(0x0000_0100_0000_01b3);
        }
    };
    for (ptr_, col, val) in parts {
        eat(ptr_.len() as u64);
        ptr_.iter().for_each(|x| eat(*x as u64));
        col.iter().for_each(|x| eat(*x as u64));
        val.iter().for_each(|x| eat(*x));
    }
    h | 1 // never 0: 0 means "nothing bound"
}

impl<E> ProvingContext<E>
where
    E: Mi355xEngine,
{
    /// Builds a new [`ProvingContext`] from `proving_key` (`ProvingContext::new`, groth16.rs:131-139).  Default: [`Self::new_in_memory`]; with the cargo
    /// feature `wire-upload`: the key travels in the reference's own wire format (`ProvingKey::serialize_unchecked`, what `ProvingContext: codec::Encode`
    /// writes, groth16.rs:166-179) through `zl_groth16_keys_from_bytes` -- one call instead of five, but it depends on a byte layout that SURVEY.md marks
    /// "[upstream], unverifiable here".
    pub fn new(proving_key: ProvingKey<E>) -> Result<Self, Error> {
        #[cfg(feature = "wire-upload")]
        {
            Self::new_from_wire(proving_key)
        }
        #[cfg(not(feature = "wire-upload"))]
        {
            Self::new_in_memory(proving_key)
        }
    }

    fn root_ctx() -> Result<*mut ffi::zl_ctx, Error> {
        let device = std::env::var("ZL_DEVICE").ok().and_then(|v| v.parse().ok()).unwrap_or(0);
        let mut p = ptr::null_mut();
        // every failure collapses into the plugin's opaque Error, like `.map_err(|_| Error)` at groth16.rs:438-465
        if unsafe { ffi::zl_ctx_create(&mut p, device) } != ffi::ZL_OK {
            return Err(Error);
        }
        Ok(p)
    }

    fn single_points(key: &ProvingKey<E>) -> [Vec<u64>; 5] {
        [E::g1_canonical(&key.vk.alpha_g1), E::g1_canonical(&key.beta_g1), E::g1_canonical(&key.delta_g1), E::g2_canonical(&key.vk.beta_g2), E::g2_canonical(&key.vk.delta_g2)]
    }

    /// The five query vectors go to the device as they lie in memory: records of `size_of::<GroupAffine>` bytes holding Montgomery coordinates and arkworks'
    /// `infinity: bool` at an offset measured on a value -- `zl_bases_upload(.., stride, inf_offset, ZL_MONT, ..)`, the layout-independent entry of the C ABI
    /// (exercised with exactly such records, G1 and G2, by tests/c/inmemory_key.c).  Keys of >= 2^19 points get the window tables the backend's own
    /// `Groth16::compile` builds (`zl_bases_precompute`, c chosen by the backend).
    pub fn new_in_memory(proving_key: ProvingKey<E>) -> Result<Self, Error> {
        let root = Self::root_ctx()?;
        let mut queries = [0u64; 5];
        let upload = |group: i32, base: *const core::ffi::c_void, n: usize, stride: usize, inf_off: usize, out: &mut u64| -> bool {
            let ok = unsafe { ffi::zl_bases_upload(root, E::CURVE, group, base, n, stride, inf_off as core::ffi::c_long, ffi::ZL_MONT, out) } == ffi::ZL_OK;
            // the tables pay for a key that is used many times and is large enough to fill the machine without them (csrc/zl_host.hip: build_window_tables)
            ok && (n < (1 << 19) || unsafe { ffi::zl_bases_precompute(root, *out, 0) } == ffi::ZL_OK)
        };
        let (s1, o1) = (core::mem::size_of::<E::G1Affine>(), E::g1_infinity_offset());
        let (s2, o2) = (core::mem::size_of::<E::G2Affine>(), E::g2_infinity_offset());
        let k = &proving_key;
        let ok = upload(ffi::ZL_G1, k.a_query.as_ptr() as *const _, k.a_query.len(), s1, o1, &mut queries[0])
            && upload(ffi::ZL_G1, k.b_g1_query.as_ptr() as *const _, k.b_g1_query.len(), s1, o1, &mut queries[1])
            && upload(ffi::ZL_G1, k.h_query.as_ptr() as *const _, k.h_query.len(), s1, o1, &mut queries[2])
            && upload(ffi::ZL_G1, k.l_query.as_ptr() as *const _, k.l_query.len(), s1, o1, &mut queries[3])
            && upload(ffi::ZL_G2, k.b_g2_query.as_ptr() as *const _, k.b_g2_query.len(), s2, o2, &mut queries[4]);
        if !ok {
            unsafe { ffi::zl_ctx_destroy(root) }; // frees whatever was uploaded
            return Err(Error);
        }
        let points = Self::single_points(&proving_key);
        Ok(Self { key: proving_key, root, queries, points, keys: ptr::null_mut(), state: Mutex::new(State { lanes: Vec::new(), r1cs: 0, shape: [0; 4], digest: 0 }), check_binding: AtomicBool::new(cfg!(debug_assertions)) })
    }

    /// The same context built through the reference's wire format (see [`Self::new`])
    pub fn new_from_wire(proving_key: ProvingKey<E>) -> Result<Self, Error> {
        let root = Self::root_ctx()?;
        let mut bytes = Vec::new();
        let mut keys = ptr::null_mut();
        let mut pk = core::mem::MaybeUninit::<ffi::zl_g16_pk>::zeroed();
        let ok = proving_key.serialize_unchecked(&mut bytes).is_ok()
            && unsafe { ffi::zl_groth16_keys_from_bytes(root, E::CURVE, bytes.as_ptr(), bytes.len(), 0, &mut keys) } == ffi::ZL_OK
            && unsafe { ffi::zl_groth16_keys_pk(keys, pk.as_mut_ptr()) } == ffi::ZL_OK;
        if !ok {
            unsafe {
                ffi::zl_groth16_keys_free(keys);
                ffi::zl_ctx_destroy(root);
            }
            return Err(Error);
        }
        let pk = unsafe { pk.assume_init() };
        let queries = [pk.a_query, pk.b_g1_query, pk.h_query, pk.l_query, pk.b_g2_query];
        let points = Self::single_points(&proving_key);
        Ok(Self { key: proving_key, root, queries, points, keys, state: Mutex::new(State { lanes: Vec::new(), r1cs: 0, shape: [0; 4], digest: 0 }), check_binding: AtomicBool::new(cfg!(debug_assertions)) })
    }

    /// `zl_g16_pk` over this context's handles and single points (valid while `self` lives)
    fn pk(&self) -> ffi::zl_g16_pk {
        ffi::zl_g16_pk {
            curve: E::CURVE,
            a_query: self.queries[0],
            b_g1_query: self.queries[1],
            h_query: self.queries[2],
            l_query: self.queries[3],
            b_g2_query: self.queries[4],
            alpha_g1: self.points[0].as_ptr(),
            beta_g1: self.points[1].as_ptr(),
            delta_g1: self.points[2].as_ptr(),
            beta_g2: self.points[3].as_ptr(),
            delta_g2: self.points[4].as_ptr(),
        }
    }

    /// an idle lane of the pool, or one more fork of the root ctx (a lane costs its own scratch buffers: the pool grows to the number of threads that
    /// ever proved at once and no further)
    fn lane(&self) -> Result<LaneGuard<'_, E>, Error> {
        let mut st = self.state.lock().map_err(|_| Error)?;
        let lane = match st.lanes.pop() {
            Some(l) => l,
            None => {
                let mut l = ptr::null_mut();
                if unsafe { ffi::zl_ctx_fork(self.root, &mut l) } != ffi::ZL_OK {
                    return Err(Error);
                }
                l
            }
        };
        Ok(LaneGuard { context: self, lane })
    }

    /// Fingerprint the compiler of every proof against the circuit whose matrices are resident (default: on in debug builds, off in release builds)
    pub fn set_check_binding(&self, on: bool) {
        self.check_binding.store(on, Ordering::Relaxed);
    }

    /// Forget the resident constraint matrices: the next proof uploads (and fingerprints) those of its own compiler.  Call it before proving ANOTHER circuit
    /// of the same shape with this key -- which only makes sense if the key was compiled for that circuit too.  The matrices are read by the lanes, and the
    /// backend refuses to free an object a live fork may read: the idle lanes are destroyed first; with a proof in flight on another thread this returns `Error`.
    pub fn rebind(&self) -> Result<(), Error> {
        let mut st = self.state.lock().map_err(|_| Error)?;
        if st.r1cs != 0 {
            for l in st.lanes.drain(..) {
                unsafe { ffi::zl_ctx_destroy(l) };
            }
            if unsafe { ffi::zl_r1cs_free(self.root, st.r1cs) } != ffi::ZL_OK {
                return Err(Error); // a lane is out proving: nothing was changed
            }
            st.r1cs = 0;
            st.shape = [0; 4];
            st.digest = 0;
        }
        Ok(())
    }
}

impl<E> Clone for ProvingContext<E>
where
    E: Mi355xEngine,
{
    /// A clone owns its own device copy of the key (and uploads the matrices again with its first proof): handles are never shared, so
    /// either side can be dropped first.  Panics if the device refuses the upload (out of memory), as `Vec::clone` would.
    fn clone(&self) -> Self {
        Self::new(self.key.clone()).expect("device upload of a cloned proving key")
    }
}

impl<E> core::fmt::Debug for ProvingContext<E>
where
    E: Mi355xEngine,
{
    fn fmt(&self, f: &mut core::fmt::Formatter<'_>) -> core::fmt::Result {
        f.debug_tuple("ProvingContext").field(&self.key).finish()
    }
}

impl<E> PartialEq for ProvingContext<E>
where
    E: Mi355xEngine,
{
    fn eq(&self, other: &Self) -> bool {
        self.key == other.key
    }
}
impl<E> Eq for ProvingContext<E> where E: Mi355xEngine {}

impl<E> CanonicalSerialize for ProvingContext<E>
where
    E: Mi355xEngine,
{
    fn serialize<W: Write>(&self, writer: W) -> Result<(), SerializationError> {
        self.key.serialize(writer)
    }
    fn serialized_size(&self) -> usize {
        self.key.serialized_size()
    }
    fn serialize_uncompressed<W: Write>(&self, writer: W) -> Result<(), SerializationError> {
        self.key.serialize_uncompressed(writer)
    }
    fn serialize_unchecked<W: Write>(&self, writer: W) -> Result<(), SerializationError> {
        self.key.serialize_unchecked(writer)
    }
    fn uncompressed_size(&self) -> usize {
        self.key.uncompressed_size()
    }
}

impl<E> CanonicalDeserialize for ProvingContext<E>
where
    E: Mi355xEngine,
{
    fn deserialize<R: Read>(reader: R) -> Result<Self, SerializationError> {
        Self::new(ProvingKey::deserialize(reader)?).map_err(|_| SerializationError::InvalidData)
    }
    fn deserialize_uncompressed<R: Read>(reader: R) -> Result<Self, SerializationError> {
        Self::new(ProvingKey::deserialize_uncompressed(reader)?).map_err(|_| SerializationError::InvalidData)
    }
    fn deserialize_unchecked<R: Read>(reader: R) -> Result<Self, SerializationError> {
        Self::new(ProvingKey::deserialize_unchecked(reader)?).map_err(|_| SerializationError::InvalidData)
    }
}

/// Same bytes as the reference's `ProvingContext: codec::Decode` (groth16.rs:142-164): `deserialize_unchecked` through `ArkReader`.
impl<E> codec::Decode for ProvingContext<E>
where
    E: Mi355xEngine,
{
    type Error = SerializationError;

    fn decode<R>(reader: R) -> Result<Self, DecodeError<R::Error, Self::Error>>
    where
        R: codec::Read,
    {
        let mut reader = ArkReader::new(reader);
        match <ProvingKey<E> as CanonicalDeserialize>::deserialize_unchecked(&mut reader) {
            Ok(key) => match reader.finish() {
                Ok(_) => Self::new(key).map_err(|_| DecodeError::Decode(SerializationError::InvalidData)),
                Err(err) => Err(DecodeError::Read(err)),
            },
            Err(err) => Err(DecodeError::Decode(err)),
        }
    }
}

/// Same bytes as the reference's `ProvingContext: codec::Encode` (groth16.rs:166-179): `serialize_unchecked` through `ArkWriter`.
impl<E> codec::Encode for ProvingContext<E>
where
    E: Mi355xEngine,
{
    fn encode<W>(&self, writer: W) -> Result<(), W::Error>
    where
        W: codec::Write,
    {
        let mut writer = ArkWriter::new(writer);
        let _ = self.key.serialize_unchecked(&mut writer);
        writer.finish().map(move |_| ())
    }
}

impl<E> Drop for ProvingContext<E>
where
    E: Mi355xEngine,
{
    fn drop(&mut self) {
        // order: the lanes (they read the root's objects), then the key object if there is one (it frees its handles on root), then the root ctx, whose
        // destruction frees every remaining device object -- the in-memory uploads and the constraint matrices (include/zl_backend.h: zl_ctx_destroy)
        unsafe {
            if let Ok(st) = self.state.get_mut() {
                for l in st.lanes.drain(..) {
                    ffi::zl_ctx_destroy(l);
                }
            }
            if !self.keys.is_null() {
                ffi::zl_groth16_keys_free(self.keys);
            }
            ffi::zl_ctx_destroy(self.root);
        }
    }
}

/// Groth16 on the MI355X backend
#[derive(Clone, Copy, Debug, Default)]
pub struct Groth16Mi355x<E>(PhantomData<E>)
where
    E: Mi355xEngine;

/// `[u64]` limbs (canonical, little-endian) -> a base-field element
fn fq_from_limbs<F: PrimeField>(limbs: &[u64]) -> F {
    let mut bytes = Vec::with_capacity(limbs.len() * 8);
    for l in limbs {
        bytes.extend_from_slice(&l.to_le_bytes());
    }
    F::from_repr(<F::BigInt as FromBytes>::read(&bytes[..]).expect("limb count matches the field")).expect("the backend returns canonical residues")
}

impl<E> Groth16Mi355x<E>
where
    E: Mi355xEngine,
    E::Fq: PrimeField,
{
    fn g1(xy: &[u64], inf: u8) -> E::G1Affine {
        if inf != 0 {
            return E::G1Affine::zero();
        }
        let n = E::FQ_LIMBS;
        E::g1_from_xy(fq_from_limbs::<E::Fq>(&xy[..n]), fq_from_limbs::<E::Fq>(&xy[n..2 * n]))
    }
}

impl<E> ProofSystem for Groth16Mi355x<E>
where
    E: Mi355xEngine,
    E::Fq: PrimeField,
{
    type Compiler = R1CS<E::Fr>;
    type PublicParameters = ();
    type ProvingContext = ProvingContext<E>;
    type VerifyingContext = VerifyingContext<E>;
    type Input = Vec<E::Fr>;
    type Proof = Proof<E>;
    type Error = Error;

    #[inline]
    fn context_compiler() -> Self::Compiler {
        Self::Compiler::for_contexts()
    }

    #[inline]
    fn proof_compiler() -> Self::Compiler {
        Self::Compiler::for_proofs()
    }

    fn compile<R>(
        public_parameters: &Self::PublicParameters,
        compiler: Self::Compiler,
        rng: &mut R,
    ) -> Result<(Self::ProvingContext, Self::VerifyingContext), Self::Error>
    where
        R: CryptoRng + RngCore + ?Sized,
    {
        let _ = public_parameters;
        // the trusted setup stays arkworks' (one-time per circuit; groth16.rs:438)
        let (key, verifying_key) = ArkGroth16::<E>::circuit_specific_setup(compiler, &mut SizedRng(rng)).map_err(|_| Error)?;
        // ... and its queries travel to the device as they lie in memory (ProvingContext::new above)
        Ok((
            ProvingContext::new(key)?,
            VerifyingContext(ArkGroth16::<E>::process_vk(&verifying_key).map_err(|_| Error)?),
        ))
    }

    fn prove<R>(context: &Self::ProvingContext, compiler: Self::Compiler, rng: &mut R) -> Result<Self::Proof, Self::Error>
    where
        R: CryptoRng + RngCore + ?Sized,
    {
        // one lane for this proof (back to the pool when `guard` drops, on every path)
        let guard = context.lane()?;
        let c = guard.lane;
        // what ark_groth16::create_proof_with_reduction does first: move the finished constraint system into a prove-mode one
        let cs = ConstraintSystem::<E::Fr>::new_ref();
        cs.set_optimization_goal(OptimizationGoal::Constraints);
        cs.set_mode(SynthesisMode::Prove { construct_matrices: true });
        compiler.generate_constraints(cs.clone()).map_err(|_| Error)?;
        cs.finalize();
        // the matrices are static per circuit: uploaded with the first proof, device-resident afterwards.  Every later compiler must have the
        // shape of the one they came from (see ProvingContext): counts are read off the constraint system, no matrix is rebuilt or hashed
        let shape = {
            let inner = cs.borrow().ok_or(Error)?;
            // (the public counters of ark-relations 0.3.0's ConstraintSystem; its a / b / c_constraints vectors are private)
            [inner.num_constraints, inner.num_instance_variables, inner.num_witness_variables, inner.num_linear_combinations]
        };
        // binding state under the mutex: the first proof of a context uploads the matrices on the ROOT ctx (single-caller: the mutex is its guard), every
        // later one only compares; two threads racing for the first proof serialise here and the second finds the matrices bound
        let mut st = context.state.lock().map_err(|_| Error)?;
        if st.r1cs != 0 && st.shape != shape {
            return Err(Error);
        }
        if st.r1cs == 0 || context.check_binding.load(Ordering::Relaxed) {
            let m = cs.to_matrices().ok_or(Error)?;
            let csr = |rows: &Vec<Vec<(E::Fr, usize)>>| {
                let (mut ptr_, mut col, mut val) = (vec![0u32], Vec::new(), Vec::<u64>::new());
                for row in rows {
                    for (coeff, index) in row {
                        col.push(*index as u32);
                        val.extend_from_slice(coeff.into_repr().as_ref()); // canonical integers (zl_r1cs: "coefficients ... canonical")
                    }
                    ptr_.push(col.len() as u32);
                }
                (ptr_, col, val)
            };
            let (a, b, cc) = (csr(&m.a), csr(&m.b), csr(&m.c));
            let fingerprint = csr_digest([(&a.0, &a.1, &a.2), (&b.0, &b.1, &b.2), (&cc.0, &cc.1, &cc.2)]);
            if st.r1cs != 0 {
                // check_binding: same shape, but is it the same circuit?
                if fingerprint != st.digest {
                    return Err(Error);
                }
            } else {
            let view = ffi::zl_r1cs {
                n_constraints: m.num_constraints as u32,
                n_instance: m.num_instance_variables as u32,
                n_witness: m.num_witness_variables as u32,
                row_ptr: [a.0.as_ptr(), b.0.as_ptr(), cc.0.as_ptr()],
                col: [a.1.as_ptr(), b.1.as_ptr(), cc.1.as_ptr()],
                val: [a.2.as_ptr(), b.2.as_ptr(), cc.2.as_ptr()],
            };
            let mut handle = 0u64;
            if unsafe { ffi::zl_r1cs_upload(context.root, E::CURVE, &view, &mut handle) } != ffi::ZL_OK {
                return Err(Error);
            }
            st.r1cs = handle;
            st.shape = shape;
            st.digest = fingerprint;
            }
        }
        let r1cs = st.r1cs;
        drop(st); // the proof itself runs outside the mutex, on this thread's lane
        // assignment = instance block (ONE, public inputs) then witnesses, as arkworks' in-memory Montgomery limbs: E::Fr is a
        // single-field tuple struct over BigInteger256, itself a single-field tuple struct over [u64; 4]; ark-ff 0.3 declares no
        // #[repr] on either, so this relies on the layout rustc gives single-field structs today (checked by the size assertion)
        assert_eq!(core::mem::size_of::<E::Fr>(), 32);
        let (instance, witness) = {
            let inner = cs.borrow().ok_or(Error)?;
            (inner.instance_assignment.clone(), inner.witness_assignment.clone())
        };
        let mut assignment = Vec::<E::Fr>::with_capacity(instance.len() + witness.len());
        assignment.extend_from_slice(&instance);
        assignment.extend_from_slice(&witness);
        // the blinding scalars, sampled exactly where create_random_proof samples them (r then s)
        let mut sized = SizedRng(rng);
        let r = E::Fr::rand(&mut sized).into_repr();
        let s = E::Fr::rand(&mut sized).into_repr();
        let pk = context.pk();
        let mut out = core::mem::MaybeUninit::<ffi::zl_g16_proof>::zeroed();
        let rc = unsafe {
            ffi::zl_groth16_prove_resident(
                c,
                &pk,
                r1cs,
                assignment.as_ptr() as *const u64,
                ffi::ZL_MONT,
                r.as_ref().as_ptr(),
                s.as_ref().as_ptr(),
                out.as_mut_ptr(),
            )
        };
        if rc != ffi::ZL_OK {
            return Err(Error); // .map_err(|_| Error) groth16.rs:456
        }
        let p = unsafe { out.assume_init() };
        let n = E::FQ_LIMBS;
        let b = if p.b_inf != 0 {
            E::G2Affine::zero()
        } else {
            // G2: x.c0 || x.c1 || y.c0 || y.c1 (zl_backend.h)
            let f = |k: usize| fq_from_limbs::<E::Fq>(&p.b[k * n..(k + 1) * n]);
            E::g2_from_xy(f(0), f(1), f(2), f(3))
        };
        Ok(Proof(ArkProof { a: Self::g1(&p.a, p.a_inf), b, c: Self::g1(&p.c, p.c_inf) }))
    }

    #[inline]
    fn verify(context: &Self::VerifyingContext, input: &Self::Input, proof: &Self::Proof) -> Result<bool, Self::Error> {
        ArkGroth16::<E>::verify_with_processed_vk(&context.0, input, &proof.0).map_err(|_| Error)
    }
}

/// Drop-in for `ark_ec::msm::VariableBaseMSM::multi_scalar_mul::<G>(bases, scalars)` against bases uploaded once with
/// [`upload_g1_bases`]: `scalars` are `into_repr()` canonical integers, passed as they lie in memory.
pub fn msm_g1<E>(bases: u64, scalars: &[<E::Fr as PrimeField>::BigInt]) -> Result<E::G1Affine, Error>
where
    E: Mi355xEngine,
    E::Fq: PrimeField,
{
    assert_eq!(core::mem::size_of::<<E::Fr as PrimeField>::BigInt>(), 32);
    let (mut xy, mut inf) = ([0u64; 12], 0u8);
    let rc = unsafe { ffi::zl_msm(ctx()?, bases, 0, scalars.as_ptr() as *const u64, scalars.len(), xy.as_mut_ptr(), &mut inf) };
    if rc != ffi::ZL_OK {
        return Err(Error);
    }
    Ok(Groth16Mi355x::<E>::g1(&xy, inf))
}

/// Uploads `bases` (arkworks' in-memory `GroupAffine { x, y, infinity }` records, Montgomery coordinates) once; the handle names them in
/// [`msm_g1`].  `infinity_offset` = byte offset of the `infinity: bool` field inside one record (`memoffset::offset_of!`).
pub fn upload_g1_bases<E>(bases: &[E::G1Affine], infinity_offset: usize) -> Result<u64, Error>
where
    E: Mi355xEngine,
{
    let mut handle = 0u64;
    let rc = unsafe {
        ffi::zl_bases_upload(
            ctx()?,
            E::CURVE,
            ffi::ZL_G1,
            bases.as_ptr() as *const core::ffi::c_void,
            bases.len(),
            core::mem::size_of::<E::G1Affine>(),
            infinity_offset as core::ffi::c_long,
            ffi::ZL_MONT,
            &mut handle,
        )
    };
    if rc == ffi::ZL_OK { Ok(handle) } else { Err(Error) }
}

/// Drop-in for `Radix2EvaluationDomain::{fft, ifft, coset_fft, coset_ifft}_in_place(&mut values)`: Montgomery limbs in place,
/// `values.len()` a power of two.
pub fn ntt_in_place<E>(values: &mut [E::Fr], inverse: bool, coset: bool) -> Result<(), Error>
where
    E: Mi355xEngine,
{
    assert!(values.len().is_power_of_two() && core::mem::size_of::<E::Fr>() == 32);
    let flags = ffi::ZL_MONT | if inverse { ffi::ZL_INVERSE } else { 0 } | if coset { ffi::ZL_COSET } else { 0 };
    let rc = unsafe { ffi::zl_ntt(ctx()?, E::CURVE, values.as_mut_ptr() as *mut u64, values.len().trailing_zeros(), flags) };
    if rc == ffi::ZL_OK { Ok(()) } else { Err(Error) }
}
