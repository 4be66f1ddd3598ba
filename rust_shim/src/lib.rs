//! Rust binding of libzkmle_amd.so (include/zkmle.h) that re-creates the reference's items for the
//! multilinear hot path.  UNCOMPILED SOURCE (no cargo/rustc in the build image): it documents the
//! binding a maintainer adds; the same ABI is exercised by the Python ctypes tests.
//!
//! Generic `F: PrimeField` cannot cross a C ABI: the shim specialises the concrete fields the reference
//! uses (ark_bls12_381::Fr, ark_bn254::Fq, ...) through `ZkField` and falls back to nothing else.
#![allow(non_camel_case_types)]
use ark_ff::PrimeField;
use std::os::raw::{c_int, c_void};

pub mod ffi {
    use super::*;
    #[repr(C)] pub struct zk_table { _p: [u8; 0] }
    #[repr(C)] pub struct zk_transcript { _p: [u8; 0] }
    #[repr(C)] pub struct zk_g1_bases { _p: [u8; 0] }
    extern "C" {
        pub fn zk_init(device: c_int) -> c_int;
        pub fn zk_status_message(status: c_int) -> *const std::os::raw::c_char;
        pub fn zk_table_upload(field: c_int, host: *const u64, len: usize, out: *mut *mut zk_table) -> c_int;
        pub fn zk_table_alloc(field: c_int, len: usize, out: *mut *mut zk_table) -> c_int;
        pub fn zk_table_download(t: *const zk_table, host: *mut u64) -> c_int;
        pub fn zk_table_free(t: *mut zk_table) -> c_int;
        pub fn zk_table_len(t: *const zk_table) -> usize;
        pub fn zk_mle_fold(inp: *const zk_table, var: usize, value: *const u64, out: *mut zk_table, stream: *mut c_void) -> c_int;
        pub fn zk_mle_evaluate(t: *const zk_table, values: *const u64, nvalues: usize, out: *mut u64) -> c_int;
        pub fn zk_mle_to_bytes(t: *const zk_table, out: *mut u8) -> c_int;
        pub fn zk_mle_scalar_mul(a: *const zk_table, s: *const u64, out: *mut zk_table, stream: *mut c_void) -> c_int;
        pub fn zk_mle_add(a: *const zk_table, b: *const zk_table, out: *mut zk_table, stream: *mut c_void) -> c_int;
        pub fn zk_mle_tensor_add(b: *const zk_table, c: *const zk_table, out: *mut zk_table, stream: *mut c_void) -> c_int;
        pub fn zk_mle_tensor_mul(b: *const zk_table, c: *const zk_table, out: *mut zk_table, stream: *mut c_void) -> c_int;
        pub fn zk_sumcheck_basic_prove(t: *const zk_table, claimed: *mut u64, rounds: *mut u64, challenges: *mut u64) -> c_int;
        pub fn zk_transcript_new(out: *mut *mut zk_transcript) -> c_int;
        pub fn zk_transcript_free(t: *mut zk_transcript) -> c_int;
        pub fn zk_transcript_append(t: *mut zk_transcript, data: *const u8, n: usize) -> c_int;
        pub fn zk_transcript_challenge(t: *mut zk_transcript, field: c_int, out: *mut u64) -> c_int;
        pub fn zk_sumcheck_gkr_prove(tables: *const *const zk_table, nprod: usize, nfac: usize, claimed: *const u64,
                                     t: *mut zk_transcript, coeffs: *mut u64, challenges: *mut u64) -> c_int;
        pub fn zk_g1_bases_upload(affine: *const u64, n: usize, out: *mut *mut zk_g1_bases) -> c_int;
        pub fn zk_g1_bases_free(b: *mut zk_g1_bases) -> c_int;
        pub fn zk_kzg_commit(poly: *const zk_table, g1: *const zk_g1_bases, out12: *mut u64) -> c_int;
        pub fn zk_kzg_open(poly: *const zk_table, g1: *const zk_g1_bases, key: *const c_void, opening: *const u64,
                           nopen: usize, n_g2: usize, evaluation: *mut u64, proofs: *mut u64) -> c_int;
        pub fn zk_kzg_setup_g1(taus: *const u64, ntaus: usize, out: *mut *mut zk_g1_bases) -> c_int;
        pub fn zk_kzg_setup_g2(taus: *const u64, ntaus: usize, out24: *mut u64) -> c_int;
        pub fn zk_kzg_verify(commitment12: *const u64, opening: *const u64, nopen: usize, evaluation: *const u64, proofs: *const u64,
                             nproofs: usize, g2_powers: *const u64, ng2: usize, ok: *mut c_int) -> c_int;
        // device-resident rounds for one-process-per-GPU provers
        pub fn zk_rounds_new(field: c_int, mode: c_int, nprod: usize, nfac: usize, nrounds: usize, t: *mut zk_transcript,
                             out: *mut *mut c_void) -> c_int;
        pub fn zk_rounds_limbs_len(r: *const c_void) -> usize;
        pub fn zk_rounds_evals(r: *mut c_void, tables: *const *const zk_table, limbs_dev: *mut u64) -> c_int;
        pub fn zk_rounds_fold_evals(r: *mut c_void, inp: *const *const zk_table, out: *const *mut zk_table, limbs_dev: *mut u64) -> c_int;
        pub fn zk_rounds_absorb(r: *mut c_void, limbs_dev: *const u64) -> c_int;
        pub fn zk_rounds_tail(r: *mut c_void, tables: *const *const zk_table) -> c_int;
        pub fn zk_rounds_collect(r: *mut c_void, t: *mut zk_transcript, claimed: *mut u64, messages: *mut u64, challenges: *mut u64,
                                 final_values: *mut u64) -> c_int;
        pub fn zk_rounds_free(r: *mut c_void) -> c_int;
    }
}

/// Concrete fields that cross the ABI.  The in-memory `Fp` is N little-endian u64 Montgomery limbs with
/// R = 2^(64 N): exactly what the library expects, so slices are passed by pointer (size asserted).
pub trait ZkField: PrimeField { const ID: c_int; const LIMBS: usize; }
impl ZkField for ark_bls12_381::Fr { const ID: c_int = 0; const LIMBS: usize = 4; }
impl ZkField for ark_bls12_381::Fq { const ID: c_int = 1; const LIMBS: usize = 6; }
impl ZkField for ark_bn254::Fq { const ID: c_int = 2; const LIMBS: usize = 4; }
impl ZkField for ark_bn254::Fr { const ID: c_int = 3; const LIMBS: usize = 4; }

fn as_limbs<F: ZkField>(v: &[F]) -> *const u64 {
    assert_eq!(std::mem::size_of::<F>(), 8 * F::LIMBS, "unexpected Fp layout");
    v.as_ptr() as *const u64
}
/// status -> the reference's behaviour: precondition codes panic with the reference's message text
fn check(rc: c_int) {
    if rc == 0 { return; }
    let msg = unsafe { std::ffi::CStr::from_ptr(ffi::zk_status_message(rc)) }.to_string_lossy().into_owned();
    panic!("{msg}");            // e.g. "Evaluated values must be a power of 2" (evaluation_form.rs:13)
}

/// polynomials::multilinear::evaluation_form::MultilinearPolynomial
#[derive(Debug, Clone, PartialEq)]
pub struct MultilinearPolynomial<F: ZkField> { pub evaluated_values: Vec<F> }

impl<F: ZkField> MultilinearPolynomial<F> {
    pub fn new(evaluated_values: &[F]) -> Self {
        assert!(evaluated_values.len().is_power_of_two(), "Evaluated values must be a power of 2");
        Self { evaluated_values: evaluated_values.to_vec() }
    }
    /// partial_evaluate (evaluation_form.rs:61): upload, one fold kernel, download.
    /// Provers should keep tables resident (`DeviceTable`) instead of paying PCIe per call.
    pub fn partial_evaluate(polynomial: &Vec<F>, evaluating_variable: usize, value: F) -> Self {
        let t = DeviceTable::<F>::upload(polynomial);
        let out = t.fold(evaluating_variable, value);
        Self { evaluated_values: out.download() }
    }
    pub fn evaluate(&self, values: &[F]) -> F {                     // :21
        let t = DeviceTable::<F>::upload(&self.evaluated_values);
        let mut out = F::zero();
        check(unsafe { ffi::zk_mle_evaluate(t.h, as_limbs(values), values.len(), &mut out as *mut F as *mut u64) });
        out
    }
    pub fn number_of_variables(&self) -> u32 { self.evaluated_values.len().ilog2() }
}

/// HBM-resident table handle (RAII)
pub struct DeviceTable<F: ZkField> { h: *mut ffi::zk_table, _f: std::marker::PhantomData<F> }
impl<F: ZkField> DeviceTable<F> {
    pub fn upload(v: &[F]) -> Self {
        let mut h = std::ptr::null_mut();
        check(unsafe { ffi::zk_table_upload(F::ID, as_limbs(v), v.len(), &mut h) });
        Self { h, _f: Default::default() }
    }
    pub fn len(&self) -> usize { unsafe { ffi::zk_table_len(self.h) } }
    pub fn fold(&self, var: usize, value: F) -> Self {
        let mut h = std::ptr::null_mut();
        check(unsafe { ffi::zk_table_alloc(F::ID, (self.len() / 2).max(1), &mut h) });
        check(unsafe { ffi::zk_mle_fold(self.h, var, &value as *const F as *const u64, h, std::ptr::null_mut()) });
        Self { h, _f: Default::default() }
    }
    pub fn download(&self) -> Vec<F> {
        let mut v = vec![F::zero(); self.len()];
        check(unsafe { ffi::zk_table_download(self.h, v.as_mut_ptr() as *mut u64) });
        v
    }
}
impl<F: ZkField> Drop for DeviceTable<F> { fn drop(&mut self) { unsafe { ffi::zk_table_free(self.h); } } }

/// sumcheck_protocol::basic_sumcheck::prover::{Prover, SumcheckProof}
pub struct SumcheckProof<F: ZkField> {
    pub initial_polynomial: MultilinearPolynomial<F>,
    pub initial_claimed_sum: F,
    pub round_univariate_polynomials: Vec<MultilinearPolynomial<F>>,
}
pub struct Prover<F: ZkField> { pub initial_polynomial: MultilinearPolynomial<F>, pub is_initialized: bool }
impl<F: ZkField> Prover<F> {
    pub fn init(values: &Vec<F>) -> Self { Self { initial_polynomial: MultilinearPolynomial::new(values), is_initialized: true } }
    pub fn prove(&mut self) -> SumcheckProof<F> {
        assert!(self.is_initialized, "Can't prove without init");
        let n = self.initial_polynomial.number_of_variables() as usize;
        let t = DeviceTable::<F>::upload(&self.initial_polynomial.evaluated_values);
        let (mut claimed, mut rounds) = (F::zero(), vec![F::zero(); 2 * n.max(1)]);
        check(unsafe { ffi::zk_sumcheck_basic_prove(t.h, &mut claimed as *mut F as *mut u64, rounds.as_mut_ptr() as *mut u64, std::ptr::null_mut()) });
        SumcheckProof {
            initial_polynomial: self.initial_polynomial.clone(), initial_claimed_sum: claimed,
            round_univariate_polynomials: rounds.chunks(2).take(n).map(MultilinearPolynomial::new).collect(),
        }
    }
}

/// multilinear_kzg::{TrustedSetup, MultilinearKZG} for P = Bls12_381, F = Fr (multilinear_kzg.rs:22-158).
/// G1 points cross the ABI as affine x || y (12 u64 Montgomery limbs, (0, 0) = infinity), G2 as x.c0 || x.c1 || y.c0 || y.c1.
pub mod kzg {
    use super::*;
    use ark_bls12_381::{Fq, Fr, G1Affine, G1Projective};
    use ark_ec::{AffineRepr, CurveGroup};

    fn g1_to_limbs(p: &G1Projective) -> [u64; 12] {
        let a = p.into_affine();
        let mut out = [0u64; 12];
        if let Some((x, y)) = a.xy() {
            out[..6].copy_from_slice(&x.0 .0);                   // Fp<_, 6>.0 = BigInt([u64; 6]), Montgomery form
            out[6..].copy_from_slice(&y.0 .0);
        }
        out
    }
    fn g1_from_limbs(l: &[u64]) -> G1Projective {
        if l.iter().all(|&w| w == 0) { return G1Projective::default(); }
        let fq = |w: &[u64]| Fq::new_unchecked(ark_ff::BigInt::new(w.try_into().unwrap()));
        G1Affine::new_unchecked(fq(&l[..6]), fq(&l[6..])).into()
    }

    pub struct TrustedSetup { pub g1: *mut ffi::zk_g1_bases, pub g2_powers: Vec<u64>, pub nvars: usize }
    impl TrustedSetup {
        pub fn initialize_setup(taus: &[Fr]) -> Self {                           // trusted_setup.rs:11-22
            let mut g1 = std::ptr::null_mut();
            check(unsafe { ffi::zk_kzg_setup_g1(as_limbs(taus), taus.len(), &mut g1) });
            let mut g2 = vec![0u64; 24 * taus.len()];
            check(unsafe { ffi::zk_kzg_setup_g2(as_limbs(taus), taus.len(), g2.as_mut_ptr()) });
            Self { g1, g2_powers: g2, nvars: taus.len() }
        }
    }
    pub struct MultilinearKZGProof { pub evaluation: Fr, pub proofs: Vec<G1Projective> }   // multilinear_kzg.rs:17-20

    pub fn commit_to_polynomial(poly: &MultilinearPolynomial<Fr>, setup: &TrustedSetup) -> G1Projective {   // :25-45
        let t = DeviceTable::<Fr>::upload(&poly.evaluated_values);
        let mut out = [0u64; 12];
        check(unsafe { ffi::zk_kzg_commit(t.h, setup.g1, out.as_mut_ptr()) });
        g1_from_limbs(&out)
    }
    pub fn open_and_prove(poly: &MultilinearPolynomial<Fr>, setup: &TrustedSetup, opening: &[Fr]) -> MultilinearKZGProof {   // :50-126
        let t = DeviceTable::<Fr>::upload(&poly.evaluated_values);
        let (mut ev, mut proofs) = (Fr::from(0u64), vec![0u64; 12 * opening.len().max(1)]);
        check(unsafe { ffi::zk_kzg_open(t.h, setup.g1, std::ptr::null(), as_limbs(opening), opening.len(), setup.nvars,
                                        &mut ev as *mut Fr as *mut u64, proofs.as_mut_ptr()) });
        MultilinearKZGProof { evaluation: ev, proofs: proofs.chunks(12).take(opening.len()).map(g1_from_limbs).collect() }
    }
    pub fn verify(setup: &TrustedSetup, commitment: &G1Projective, opening: &[Fr], proof: &MultilinearKZGProof) -> bool {    // :131-158
        let c = g1_to_limbs(commitment);
        let prs: Vec<u64> = proof.proofs.iter().flat_map(|p| g1_to_limbs(p)).collect();
        let mut ok: c_int = 0;
        check(unsafe { ffi::zk_kzg_verify(c.as_ptr(), as_limbs(opening), opening.len(), &proof.evaluation as *const Fr as *const u64,
                                          prs.as_ptr(), proof.proofs.len(), setup.g2_powers.as_ptr(), setup.nvars, &mut ok) });
        ok == 1
    }
}
// sumcheck_gkr_protocol::prove and gkr_protocol::prove bind zk_sumcheck_gkr_prove / zk_gkr_prove the same way.
