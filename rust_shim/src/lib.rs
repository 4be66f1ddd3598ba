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
    #[repr(C)] #[derive(Clone, Copy)] pub struct zk_gate { pub left: u64, pub right: u64, pub out: u64, pub op: u64 }   // op 0 = Add, 1 = Mul
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
        pub fn zk_sumcheck_gkr_verify(field: c_int, claimed: *const u64, coeffs: *const u64, nrounds: usize, ncoef: usize,
                                      t: *mut zk_transcript, challenges: *mut u64, last_claimed_sum: *mut u64, ok: *mut c_int) -> c_int;
        pub fn zk_sumpoly_evaluate(tables: *const *const zk_table, nprod: usize, nfac: usize, values: *const u64, nvalues: usize,
                                   out: *mut u64) -> c_int;
        pub fn zk_sumpoly_reduce(tables: *const *const zk_table, nprod: usize, nfac: usize, out: *mut zk_table) -> c_int;
        pub fn zk_prodpoly_reduce(tables: *const *const zk_table, nfac: usize, out: *mut zk_table) -> c_int;
        pub fn zk_sumpoly_round_evals(tables: *const *const zk_table, nprod: usize, nfac: usize, out: *mut u64) -> c_int;
        pub fn zk_gkr_rounds(layer_index: usize) -> usize;
        pub fn zk_gkr_prove(field: c_int, gates: *const zk_gate, gate_counts: *const usize, nlayers: usize, inputs: *const u64,
                            ninputs: usize, circuit_output: *mut u64, output_len: *mut usize, claimed_sum: *mut u64,
                            layer_claims: *mut u64, coeffs: *mut u64, challenges: *mut u64, wb_evals: *mut u64, wc_evals: *mut u64) -> c_int;
        pub fn zk_gkr_verify(field: c_int, gates: *const zk_gate, gate_counts: *const usize, nlayers: usize, inputs: *const u64,
                             ninputs: usize, circuit_output: *const u64, output_len: usize, layer_claims: *const u64,
                             coeffs: *const u64, wb_evals: *const u64, wc_evals: *const u64, ok: *mut c_int) -> c_int;
        pub fn zk_gkr_prove_succinct(gates: *const zk_gate, gate_counts: *const usize, nlayers: usize, inputs: *const u64, ninputs: usize,
                                     g1_powers: *const zk_g1_bases, n_g2: usize, circuit_output: *mut u64, output_len: *mut usize,
                                     claimed_sum: *mut u64, layer_claims: *mut u64, coeffs: *mut u64, challenges: *mut u64,
                                     wb_evals: *mut u64, wc_evals: *mut u64, commitment12: *mut u64, rb_evaluation: *mut u64,
                                     rb_proofs: *mut u64, rc_evaluation: *mut u64, rc_proofs: *mut u64) -> c_int;
        pub fn zk_gkr_verify_succinct(gates: *const zk_gate, gate_counts: *const usize, nlayers: usize, circuit_output: *const u64,
                                      output_len: usize, layer_claims: *const u64, coeffs: *const u64, wb_evals: *const u64,
                                      wc_evals: *const u64, commitment12: *const u64, rb_evaluation: *const u64, rb_proofs: *const u64,
                                      n_rb_proofs: usize, rc_evaluation: *const u64, rc_proofs: *const u64, n_rc_proofs: usize,
                                      g2_powers: *const u64, ng2: usize, ok: *mut c_int) -> c_int;
        // multi-GPU provers: one process per GPU, RCCL inside the library (include/zkmle.h zk_comm_*, zk_sharded_*)
        pub fn zk_comm_unique_id(out128: *mut u8) -> c_int;
        pub fn zk_comm_init_rccl(id128: *const u8, nranks: c_int, rank: c_int, out: *mut *mut c_void) -> c_int;
        pub fn zk_comm_free(c: *mut c_void) -> c_int;
        // the ranks as threads of this process (one thread per GPU): include/zkmle.h zk_comm_local_group_*
        pub fn zk_comm_local_group_new(nranks: c_int, out: *mut *mut c_void) -> c_int;
        pub fn zk_comm_local_group_free(g: *mut c_void) -> c_int;
        pub fn zk_comm_local_group_abort(g: *mut c_void) -> c_int;
        pub fn zk_comm_from_local_group(g: *mut c_void, rank: c_int, out: *mut *mut c_void) -> c_int;
        // once per TrustedSetup: window-shifted copies of g1_powers_of_tau, so that every later commit uses one bucket set (22-bit windows)
        pub fn zk_g1_bases_precompute(b: *mut zk_g1_bases, window_bits: c_int) -> c_int;
        pub fn zk_kzg_opening_key_precompute(key: *mut c_void, window_bits: c_int, min_points: usize) -> c_int;   // key: zk_kzg_opening_key_new
        pub fn zk_sharded_sumcheck_basic_prove(c: *mut c_void, shard: *const zk_table, absorb_table: c_int, claimed: *mut u64,
                                               rounds: *mut u64, challenges: *mut u64) -> c_int;
        pub fn zk_sharded_sumcheck_gkr_prove(c: *mut c_void, shards: *const *const zk_table, nprod: usize, nfac: usize, claimed: *const u64,
                                             t: *mut zk_transcript, coeffs: *mut u64, challenges: *mut u64, final_values: *mut u64) -> c_int;
        pub fn zk_sharded_msm_g1(c: *mut c_void, scalars: *const zk_table, bases: *const zk_g1_bases, window_bits: c_int, out12: *mut u64,
                                 stats: *mut c_void) -> c_int;
        pub fn zk_sharded_kzg_open(c: *mut c_void, shard: *const zk_table, bases_local: *const zk_g1_bases, key_local: *const c_void,
                                   opening: *const u64, nopen: usize, evaluation: *mut u64, proofs: *mut u64) -> c_int;
        // device-resident rounds for one-process-per-GPU provers
        pub fn zk_rounds_new(field: c_int, mode: c_int, nprod: usize, nfac: usize, nrounds: usize, t: *mut zk_transcript,
                             out: *mut *mut c_void) -> c_int;
        pub fn zk_rounds_limbs_len(r: *const c_void) -> usize;
        pub fn zk_rounds_evals(r: *mut c_void, tables: *const *const zk_table, limbs_dev: *mut u64) -> c_int;
        pub fn zk_rounds_fold_evals(r: *mut c_void, inp: *const *const zk_table, out: *const *mut zk_table, limbs_dev: *mut u64) -> c_int;
        pub fn zk_rounds_absorb(r: *mut c_void, limbs_dev: *const u64) -> c_int;
        pub fn zk_rounds_tail(r: *mut c_void, tables: *const *const zk_table) -> c_int;
        // basic sumcheck, several rounds per pass and per all-reduce (include/zkmle.h)
        pub fn zk_rounds_multi_max(r: *const c_void) -> u32;
        pub fn zk_rounds_multi_evals(r: *mut c_void, table: *const zk_table, m: u32, limbs_dev: *mut u64) -> c_int;
        pub fn zk_rounds_multi_absorb(r: *mut c_void, limbs_dev: *const u64, m: u32) -> c_int;
        pub fn zk_rounds_multi_fold_evals(r: *mut c_void, inp: *const zk_table, out: *mut zk_table, k: u32, m_next: u32, limbs_dev: *mut u64) -> c_int;
        pub fn zk_rounds_multi_tail(r: *mut c_void, table: *const zk_table) -> c_int;
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

    pub(crate) fn g1_to_limbs(p: &G1Projective) -> [u64; 12] {
        let a = p.into_affine();
        let mut out = [0u64; 12];
        if let Some((x, y)) = a.xy() {
            out[..6].copy_from_slice(&x.0 .0);                   // Fp<_, 6>.0 = BigInt([u64; 6]), Montgomery form
            out[6..].copy_from_slice(&y.0 .0);
        }
        out
    }
    pub(crate) fn g1_from_limbs(l: &[u64]) -> G1Projective {
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
        /// Optional, once per setup: window-shifted copies of the G1 powers (ceil(256 / 22) x 128 bytes per point of HBM), after which every
        /// commit_to_polynomial against this setup uses 22-bit windows on one bucket set.  The commitments are the same group elements.
        pub fn precompute_for_commits(&mut self) { check(unsafe { ffi::zk_g1_bases_precompute(self.g1, 0) }); }
    }
    #[derive(Clone)] pub struct MultilinearKZGProof { pub evaluation: Fr, pub proofs: Vec<G1Projective> }   // multilinear_kzg.rs:17-20

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

// =====================================================================================================================
// The GKR half of the path: composed polynomials, the degree-2 sumcheck, gkr_protocol and succinct_gkr_protocol.
// Same item names, field names and signatures as the reference; every prover is ONE call into the library.
// =====================================================================================================================

/// transcripts::fiat_shamir::fiat_shamir_transcript::Transcript (fiat_shamir_transcript.rs:5-43): the sponge lives in the
/// library (host Keccak-256 for the big absorbs, handed to the device for the rounds of a sumcheck).
pub struct Transcript { h: *mut ffi::zk_transcript }
impl Transcript {
    pub fn new() -> Self {                                                      // :12
        let mut h = std::ptr::null_mut();
        check(unsafe { ffi::zk_transcript_new(&mut h) });
        Self { h }
    }
    pub fn append(&mut self, incoming_data: &[u8]) {                            // :22
        check(unsafe { ffi::zk_transcript_append(self.h, incoming_data.as_ptr(), incoming_data.len()) });
    }
    pub fn random_challenge_as_field_element<F: ZkField>(&mut self) -> F {      // :38
        let mut out = F::zero();
        check(unsafe { ffi::zk_transcript_challenge(self.h, F::ID, &mut out as *mut F as *mut u64) });
        out
    }
}
impl Drop for Transcript { fn drop(&mut self) { unsafe { ffi::zk_transcript_free(self.h); } } }

/// polynomials::composed::product_polynomial::ProductPolynomial (product_polynomial.rs:6-8)
#[derive(Clone, Debug, PartialEq)]
pub struct ProductPolynomial<F: ZkField> { pub polynomials: Vec<MultilinearPolynomial<F>> }
impl<F: ZkField> ProductPolynomial<F> {
    pub fn new(polynomials: Vec<MultilinearPolynomial<F>>) -> Self {            // :11-24
        let n = polynomials[0].number_of_variables();
        assert!(polynomials.iter().all(|p| p.number_of_variables() == n), "different number of variables");
        Self { polynomials }
    }
    pub fn evaluate(&self, values: &Vec<F>) -> F {                              // :26-34
        self.polynomials.iter().fold(F::one(), |acc, p| acc * p.evaluate(values))
    }
    pub fn partial_evaluate(&self, evaluating_variable: usize, value: F) -> Self {   // :36-54
        Self { polynomials: self.polynomials.iter()
            .map(|p| MultilinearPolynomial::partial_evaluate(&p.evaluated_values, evaluating_variable, value)).collect() }
    }
    pub fn multiply_polynomials_element_wise(&self) -> MultilinearPolynomial<F> {    // :58-73
        assert!(self.polynomials.len() > 1, "more than one polynomial required for mul operation");
        let d = DeviceSum::<F>::upload(&[self.polynomials.iter().collect::<Vec<_>>()]);
        let out = DeviceTable::<F>::alloc(d.tabs[0].len());
        check(unsafe { ffi::zk_prodpoly_reduce(d.ptrs.as_ptr(), d.nfac, out.h) });
        MultilinearPolynomial { evaluated_values: out.download() }
    }
    pub fn degree(&self) -> usize { self.polynomials.len() }                    // :85-87
}

/// polynomials::composed::sum_polynomial::SumPolynomial (sum_polynomial.rs:7-9)
#[derive(Clone, Debug, PartialEq)]
pub struct SumPolynomial<F: ZkField> { pub product_polynomials: Vec<ProductPolynomial<F>> }

/// every table of a SumPolynomial resident in HBM, product-major ([p * nfac + f]), as the library's array-of-handles argument
struct DeviceSum<F: ZkField> { tabs: Vec<DeviceTable<F>>, ptrs: Vec<*const ffi::zk_table>, nprod: usize, nfac: usize }
impl<F: ZkField> DeviceSum<F> {
    fn upload(products: &[Vec<&MultilinearPolynomial<F>>]) -> Self {
        let (nprod, nfac) = (products.len(), products[0].len());
        let tabs: Vec<_> = products.iter().flat_map(|p| p.iter().map(|t| DeviceTable::<F>::upload(&t.evaluated_values))).collect();
        let ptrs = tabs.iter().map(|t| t.h as *const ffi::zk_table).collect();
        Self { tabs, ptrs, nprod, nfac }
    }
}
impl<F: ZkField> SumPolynomial<F> {
    pub fn new(product_polynomials: Vec<ProductPolynomial<F>>) -> Self {        // :12-28
        let n = product_polynomials[0].polynomials[0].number_of_variables();
        assert!(product_polynomials.iter().all(|pp| pp.polynomials.iter().all(|p| p.number_of_variables() == n)),
                "different number of variables");
        Self { product_polynomials }
    }
    fn table_refs(&self) -> Vec<Vec<&MultilinearPolynomial<F>>> {
        self.product_polynomials.iter().map(|pp| pp.polynomials.iter().collect()).collect()
    }
    fn reduce_tables(products: &[Vec<&MultilinearPolynomial<F>>]) -> MultilinearPolynomial<F> {
        let d = DeviceSum::<F>::upload(products);
        let out = DeviceTable::<F>::alloc(d.tabs[0].len());
        check(unsafe { ffi::zk_sumpoly_reduce(d.ptrs.as_ptr(), d.nprod, d.nfac, out.h) });
        MultilinearPolynomial { evaluated_values: out.download() }
    }
    pub fn evaluate(&self, values: &Vec<F>) -> F {                              // :30-38
        let d = DeviceSum::<F>::upload(&self.table_refs());
        let mut out = F::zero();
        check(unsafe { ffi::zk_sumpoly_evaluate(d.ptrs.as_ptr(), d.nprod, d.nfac, as_limbs(values), values.len(), &mut out as *mut F as *mut u64) });
        out
    }
    pub fn partial_evaluate(&self, evaluating_variable: usize, value: F) -> Self {   // :40-53
        Self { product_polynomials: self.product_polynomials.iter().map(|pp| pp.partial_evaluate(evaluating_variable, value)).collect() }
    }
    pub fn add_polynomials_element_wise(&self) -> MultilinearPolynomial<F> {    // :57-76
        assert!(self.product_polynomials.len() > 1, "more than one product polynomial required for add operation");
        Self::reduce_tables(&self.table_refs())
    }
    pub fn degree(&self) -> usize { self.product_polynomials[0].degree() }      // :88-90
    pub fn number_of_variables(&self) -> u32 { self.product_polynomials[0].polynomials[0].number_of_variables() }   // :92-94
}
impl<F: ZkField> DeviceTable<F> {
    pub fn alloc(len: usize) -> Self {
        let mut h = std::ptr::null_mut();
        check(unsafe { ffi::zk_table_alloc(F::ID, len, &mut h) });
        Self { h, _f: Default::default() }
    }
}

/// polynomials::univariate::dense_univariate::DenseUnivariatePolynomial: only what a round polynomial needs
#[derive(Clone, Debug, PartialEq)]
pub struct DenseUnivariatePolynomial<F: ZkField> { pub coefficients: Vec<F> }
impl<F: ZkField> DenseUnivariatePolynomial<F> {
    pub fn new(coefficients: Vec<F>) -> Self { Self { coefficients } }
    pub fn evaluate(&self, x: F) -> F { self.coefficients.iter().rev().fold(F::zero(), |acc, c| acc * x + c) }   // dense_univariate.rs:57-68
}

/// sumcheck_protocol::gkr_sumcheck::sumcheck_gkr_protocol (sumcheck_gkr_protocol.rs:8-21)
pub mod sumcheck_gkr_protocol {
    use super::*;
    #[derive(Clone, Debug)]
    pub struct SumcheckProverProof<F: ZkField> {
        pub claimed_sum: F,
        pub round_univariate_polynomials: Vec<DenseUnivariatePolynomial<F>>,
        pub random_challenges: Vec<F>,
    }
    #[derive(Clone, Debug)]
    pub struct SumcheckVerifierProof<F: ZkField> { pub is_proof_valid: bool, pub random_challenges: Vec<F>, pub last_claimed_sum: F }

    /// prove (:24-67): every round -- fused fold + evaluations, reduction, interpolation, absorb, challenge -- runs on the
    /// device; one synchronisation.  `transcript` continues from the caller's state and is advanced as the reference's is.
    pub fn prove<F: ZkField>(sum_polynomial: SumPolynomial<F>, claimed_sum: F, transcript: &mut Transcript) -> SumcheckProverProof<F> {
        let n = sum_polynomial.number_of_variables() as usize;
        let ncoef = sum_polynomial.degree() + 1;
        let d = DeviceSum::<F>::upload(&sum_polynomial.table_refs());
        let (mut coeffs, mut chal) = (vec![F::zero(); (n * ncoef).max(1)], vec![F::zero(); n.max(1)]);
        check(unsafe { ffi::zk_sumcheck_gkr_prove(d.ptrs.as_ptr(), d.nprod, d.nfac, &claimed_sum as *const F as *const u64, transcript.h,
                                                  coeffs.as_mut_ptr() as *mut u64, chal.as_mut_ptr() as *mut u64) });
        SumcheckProverProof {
            claimed_sum,
            round_univariate_polynomials: coeffs.chunks(ncoef).take(n).map(|c| DenseUnivariatePolynomial::new(c.to_vec())).collect(),
            random_challenges: chal[..n].to_vec(),
        }
    }
    /// verify (:69-111)
    pub fn verify<F: ZkField>(proof: &SumcheckProverProof<F>, transcript: &mut Transcript) -> SumcheckVerifierProof<F> {
        let n = proof.round_univariate_polynomials.len();
        let ncoef = proof.round_univariate_polynomials.first().map_or(0, |p| p.coefficients.len());
        let flat: Vec<F> = proof.round_univariate_polynomials.iter().flat_map(|p| p.coefficients.iter().cloned()).collect();
        let (mut chal, mut last, mut ok) = (vec![F::zero(); n.max(1)], F::zero(), 0 as c_int);
        check(unsafe { ffi::zk_sumcheck_gkr_verify(F::ID, &proof.claimed_sum as *const F as *const u64, as_limbs(&flat), n, ncoef, transcript.h,
                                                   chal.as_mut_ptr() as *mut u64, &mut last as *mut F as *mut u64, &mut ok) });
        SumcheckVerifierProof { is_proof_valid: ok == 1, random_challenges: chal[..n].to_vec(), last_claimed_sum: last }
    }
    /// generate_round_univariate (:113-143): the evaluations at 0..=degree of the current round
    pub fn generate_round_univariate<F: ZkField>(sum_polynomial: &SumPolynomial<F>) -> Vec<F> {
        let d = DeviceSum::<F>::upload(&sum_polynomial.table_refs());
        let mut out = vec![F::zero(); sum_polynomial.degree() + 1];
        check(unsafe { ffi::zk_sumpoly_round_evals(d.ptrs.as_ptr(), d.nprod, d.nfac, out.as_mut_ptr() as *mut u64) });
        out
    }
}

/// circuit::arithmetic_circuit::{Operator, Gate, Layer, Circuit} (arithmetic_circuit.rs:5-30)
pub mod circuit {
    use super::*;
    pub enum Operator { Add, Mul }
    pub struct Gate { pub left_index: usize, pub right_index: usize, pub output_index: usize, pub operator: Operator }
    impl Gate {
        pub fn new(left_index: usize, right_index: usize, output_index: usize, operator: Operator) -> Self {
            Self { left_index, right_index, output_index, operator }
        }
    }
    pub struct Layer { pub gates: Vec<Gate> }
    impl Layer { pub fn new(gates: Vec<Gate>) -> Self { Self { gates } } }
    pub struct Circuit<F: ZkField> { pub layers: Vec<Layer>, _phantom: std::marker::PhantomData<F> }
    impl<F: ZkField> Circuit<F> {
        pub fn new(layers: Vec<Layer>) -> Self { Self { layers, _phantom: Default::default() } }
        /// the library's flat form: all gates, layer 0 (the output layer) first, and the per-layer counts
        pub(crate) fn flat(&self) -> (Vec<ffi::zk_gate>, Vec<usize>) {
            let gates = self.layers.iter().flat_map(|l| l.gates.iter().map(|g| ffi::zk_gate {
                left: g.left_index as u64, right: g.right_index as u64, out: g.output_index as u64,
                op: match g.operator { Operator::Add => 0, Operator::Mul => 1 } })).collect();
            (gates, self.layers.iter().map(|l| l.gates.len()).collect())
        }
    }
}

/// flattened proof buffers of zk_gkr_prove / zk_gkr_prove_succinct and their (un)packing into the reference's structs
struct FlatGkr<F: ZkField> { out: Vec<F>, out_len: usize, claimed: F, claims: Vec<F>, coeffs: Vec<F>, chal: Vec<F>, wb: Vec<F>, wc: Vec<F> }
impl<F: ZkField> FlatGkr<F> {
    fn rounds(nlayers: usize) -> Vec<usize> { (0..nlayers).map(|l| unsafe { ffi::zk_gkr_rounds(l) }).collect() }
    fn alloc(nlayers: usize) -> Self {
        let total: usize = Self::rounds(nlayers).iter().sum();
        Self { out: vec![F::zero(); 1usize << nlayers.max(1)], out_len: 0, claimed: F::zero(), claims: vec![F::zero(); nlayers],
               coeffs: vec![F::zero(); 3 * total], chal: vec![F::zero(); total], wb: vec![F::zero(); nlayers.max(1)], wc: vec![F::zero(); nlayers.max(1)] }
    }
    fn sumcheck_proofs(&self, nlayers: usize) -> Vec<sumcheck_gkr_protocol::SumcheckProverProof<F>> {
        let mut off = 0;
        Self::rounds(nlayers).iter().enumerate().map(|(l, &r)| {
            let p = sumcheck_gkr_protocol::SumcheckProverProof {
                claimed_sum: self.claims[l],
                round_univariate_polynomials: self.coeffs[3 * off..3 * (off + r)].chunks(3).map(|c| DenseUnivariatePolynomial::new(c.to_vec())).collect(),
                random_challenges: self.chal[off..off + r].to_vec(),
            };
            off += r;
            p
        }).collect()
    }
    fn from_proofs(proofs: &[sumcheck_gkr_protocol::SumcheckProverProof<F>]) -> (Vec<F>, Vec<F>) {
        (proofs.iter().map(|p| p.claimed_sum).collect(),
         proofs.iter().flat_map(|p| p.round_univariate_polynomials.iter().flat_map(|u| u.coefficients.iter().cloned())).collect())
    }
}

/// gkr::gkr_protocol (gkr_protocol.rs:17-23, :26, :146)
pub mod gkr_protocol {
    use super::*;
    use super::circuit::Circuit;
    use super::sumcheck_gkr_protocol::SumcheckProverProof;
    #[derive(Clone, Debug)]
    pub struct Proof<F: ZkField> {
        pub circuit_output: Vec<F>,
        pub claimed_sum: F,
        pub sumcheck_proofs: Vec<SumcheckProverProof<F>>,
        pub wb_evaluations: Vec<F>,
        pub wc_evaluations: Vec<F>,
    }
    /// prove (:26-143): circuit evaluation, wiring predicates, alpha/beta folding, f(b,c), every layer's sumcheck, wb / wc
    pub fn prove<F: ZkField>(circuit: &mut Circuit<F>, inputs: &[F]) -> Proof<F> {
        let (gates, counts) = circuit.flat();
        let nl = counts.len();
        let mut b = FlatGkr::<F>::alloc(nl);
        check(unsafe { ffi::zk_gkr_prove(F::ID, gates.as_ptr(), counts.as_ptr(), nl, as_limbs(inputs), inputs.len(),
                                         b.out.as_mut_ptr() as *mut u64, &mut b.out_len, &mut b.claimed as *mut F as *mut u64,
                                         b.claims.as_mut_ptr() as *mut u64, b.coeffs.as_mut_ptr() as *mut u64, b.chal.as_mut_ptr() as *mut u64,
                                         b.wb.as_mut_ptr() as *mut u64, b.wc.as_mut_ptr() as *mut u64) });
        Proof { circuit_output: b.out[..b.out_len].to_vec(), claimed_sum: b.claimed, sumcheck_proofs: b.sumcheck_proofs(nl),
                wb_evaluations: b.wb[..nl - 1].to_vec(), wc_evaluations: b.wc[..nl - 1].to_vec() }
    }
    /// verify (:146-236)
    pub fn verify<F: ZkField>(circuit: &mut Circuit<F>, proof: Proof<F>, inputs: &[F]) -> bool {
        let (gates, counts) = circuit.flat();
        let (claims, coeffs) = FlatGkr::<F>::from_proofs(&proof.sumcheck_proofs);
        let mut ok: c_int = 0;
        check(unsafe { ffi::zk_gkr_verify(F::ID, gates.as_ptr(), counts.as_ptr(), counts.len(), as_limbs(inputs), inputs.len(),
                                          as_limbs(&proof.circuit_output), proof.circuit_output.len(), as_limbs(&claims), as_limbs(&coeffs),
                                          as_limbs(&proof.wb_evaluations), as_limbs(&proof.wc_evaluations), &mut ok) });
        ok == 1
    }
}

/// gkr::succinct_gkr_protocol for P = Bls12_381, F = Fr (succinct_gkr_protocol.rs:23-32, :35, :172)
pub mod succinct_gkr_protocol {
    use super::*;
    use super::circuit::Circuit;
    use super::kzg::{MultilinearKZGProof, TrustedSetup};
    use super::sumcheck_gkr_protocol::SumcheckProverProof;
    use ark_bls12_381::{Fr, G1Projective};
    #[derive(Clone)]
    pub struct SuccinctProof {
        pub circuit_output: Vec<Fr>,
        pub claimed_sum: Fr,
        pub sumcheck_proofs: Vec<SumcheckProverProof<Fr>>,
        pub wb_evaluations: Vec<Fr>,
        pub wc_evaluations: Vec<Fr>,
        pub input_polynomial_commitment: G1Projective,
        pub input_rb_proof: MultilinearKZGProof,
        pub input_rc_proof: MultilinearKZGProof,
    }
    /// prove_succinct (:35-169): the GKR proof + commit(inputs) (:42-44) + open(inputs, rb), open(inputs, rc) (:154-157)
    pub fn prove_succinct(circuit: &mut Circuit<Fr>, inputs: &[Fr], trusted_setup: &TrustedSetup) -> SuccinctProof {
        let (gates, counts) = circuit.flat();
        let nl = counts.len();
        let mut b = FlatGkr::<Fr>::alloc(nl);
        let (mut com, mut rb_ev, mut rc_ev) = ([0u64; 12], Fr::from(0u64), Fr::from(0u64));
        let (mut rb, mut rc) = (vec![0u64; 12 * nl], vec![0u64; 12 * nl]);
        check(unsafe { ffi::zk_gkr_prove_succinct(gates.as_ptr(), counts.as_ptr(), nl, as_limbs(inputs), inputs.len(), trusted_setup.g1,
                                                  trusted_setup.nvars, b.out.as_mut_ptr() as *mut u64, &mut b.out_len,
                                                  &mut b.claimed as *mut Fr as *mut u64, b.claims.as_mut_ptr() as *mut u64,
                                                  b.coeffs.as_mut_ptr() as *mut u64, b.chal.as_mut_ptr() as *mut u64,
                                                  b.wb.as_mut_ptr() as *mut u64, b.wc.as_mut_ptr() as *mut u64, com.as_mut_ptr(),
                                                  &mut rb_ev as *mut Fr as *mut u64, rb.as_mut_ptr(), &mut rc_ev as *mut Fr as *mut u64, rc.as_mut_ptr()) });
        let pts = |v: &[u64]| v.chunks(12).map(kzg::g1_from_limbs).collect::<Vec<_>>();
        SuccinctProof { circuit_output: b.out[..b.out_len].to_vec(), claimed_sum: b.claimed, sumcheck_proofs: b.sumcheck_proofs(nl),
                        wb_evaluations: b.wb[..nl - 1].to_vec(), wc_evaluations: b.wc[..nl - 1].to_vec(),
                        input_polynomial_commitment: kzg::g1_from_limbs(&com),
                        input_rb_proof: MultilinearKZGProof { evaluation: rb_ev, proofs: pts(&rb) },
                        input_rc_proof: MultilinearKZGProof { evaluation: rc_ev, proofs: pts(&rc) } }
    }
    /// verify_succinct (:172-285): GKR rounds + two KZG verifications (n + 1 pairings each, host side of the library)
    pub fn verify_succinct(circuit: &mut Circuit<Fr>, proof: SuccinctProof, trusted_setup: &TrustedSetup) -> bool {
        let (gates, counts) = circuit.flat();
        let (claims, coeffs) = FlatGkr::<Fr>::from_proofs(&proof.sumcheck_proofs);
        let flat = |p: &MultilinearKZGProof| p.proofs.iter().flat_map(|q| kzg::g1_to_limbs(q)).collect::<Vec<u64>>();
        let (com, rb, rc) = (kzg::g1_to_limbs(&proof.input_polynomial_commitment), flat(&proof.input_rb_proof), flat(&proof.input_rc_proof));
        let mut ok: c_int = 0;
        check(unsafe { ffi::zk_gkr_verify_succinct(gates.as_ptr(), counts.as_ptr(), counts.len(), as_limbs(&proof.circuit_output),
                                                   proof.circuit_output.len(), as_limbs(&claims), as_limbs(&coeffs),
                                                   as_limbs(&proof.wb_evaluations), as_limbs(&proof.wc_evaluations), com.as_ptr(),
                                                   &proof.input_rb_proof.evaluation as *const Fr as *const u64, rb.as_ptr(), proof.input_rb_proof.proofs.len(),
                                                   &proof.input_rc_proof.evaluation as *const Fr as *const u64, rc.as_ptr(), proof.input_rc_proof.proofs.len(),
                                                   trusted_setup.g2_powers.as_ptr(), trusted_setup.nvars, &mut ok) });
        ok == 1
    }
}
