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
        pub fn zk_sumcheck_basic_prove_on(t: *const zk_table, transcript: *mut zk_transcript, claimed: *mut u64, rounds: *mut u64, challenges: *mut u64) -> c_int;
        pub fn zk_sumcheck_basic_verify(t: *const zk_table, claimed: *const u64, rounds: *const u64, nrounds: usize, ok: *mut c_int) -> c_int;
        pub fn zk_mle_sum(t: *const zk_table, out: *mut u64) -> c_int;
        pub fn zk_mle_half_sums(t: *const zk_table, out2: *mut u64) -> c_int;
        pub fn zk_fe_to_bytes_be(field: c_int, a: *const u64, out: *mut u8) -> c_int;
        pub fn zk_vec_to_canonical(field: c_int, mont: *const u64, n: usize, canon: *mut u64) -> c_int;
        pub fn zk_uni_evaluate(field: c_int, coeffs: *const u64, n: usize, x: *const u64, out: *mut u64) -> c_int;
        pub fn zk_uni_lagrange_interpolate(field: c_int, xs: *const u64, ys: *const u64, n: usize, out: *mut u64) -> c_int;
        pub fn zk_transcript_sample(t: *mut zk_transcript, out32: *mut u8) -> c_int;
        pub fn zk_num_of_layer_variables(layer_index: usize) -> usize;
        pub fn zk_wiring_index(layer_index: usize, a: usize, b: usize, c: usize) -> usize;
        pub fn zk_circuit_eval_size(gates: *const zk_gate, gate_counts: *const usize, nlayers: usize, ninputs: usize) -> usize;
        pub fn zk_circuit_evaluate(field: c_int, gates: *const zk_gate, gate_counts: *const usize, nlayers: usize, inputs: *const u64,
                                   ninputs: usize, layer_sizes: *mut usize, evals: *mut u64) -> c_int;
        pub fn zk_circuit_add_mul_mle(field: c_int, layer_gates: *const zk_gate, ngates: usize, layer_index: usize,
                                      add_i: *mut *mut zk_table, mul_i: *mut *mut zk_table) -> c_int;
        pub fn zk_g1_bases_download(b: *const zk_g1_bases, affine: *mut u64) -> c_int;
        pub fn zk_g1_bases_len(b: *const zk_g1_bases) -> usize;
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

/// The pairing the KZG half of the path runs on (the reference instantiates `P = Bls12_381` everywhere: multilinear_kzg.rs:214,
/// succinct_gkr_protocol.rs:293).  G1 points cross the ABI as affine x || y (12 u64 Montgomery limbs, all zero = infinity), G2 points as
/// x.c0 || x.c1 || y.c0 || y.c1 (24 limbs).
pub trait ZkPairing: ark_ec::pairing::Pairing {
    fn g1_to_limbs(p: &Self::G1) -> [u64; 12];
    fn g1_from_limbs(l: &[u64]) -> Self::G1;
    fn g2_to_limbs(p: &Self::G2) -> [u64; 24];
    fn g2_from_limbs(l: &[u64]) -> Self::G2;
}
impl ZkPairing for ark_bls12_381::Bls12_381 {
    fn g1_to_limbs(p: &Self::G1) -> [u64; 12] {
        use ark_ec::{AffineRepr, CurveGroup};
        let mut out = [0u64; 12];
        if let Some((x, y)) = p.into_affine().xy() {
            out[..6].copy_from_slice(&x.0 .0);                   // Fp<_, 6>.0 = BigInt([u64; 6]), Montgomery form
            out[6..].copy_from_slice(&y.0 .0);
        }
        out
    }
    fn g1_from_limbs(l: &[u64]) -> Self::G1 {
        use ark_bls12_381::{Fq, G1Affine};
        if l.iter().all(|&w| w == 0) { return Self::G1::default(); }
        let fq = |w: &[u64]| Fq::new_unchecked(ark_ff::BigInt::new(w.try_into().unwrap()));
        G1Affine::new_unchecked(fq(&l[..6]), fq(&l[6..])).into()
    }
    fn g2_to_limbs(p: &Self::G2) -> [u64; 24] {
        use ark_ec::{AffineRepr, CurveGroup};
        let mut out = [0u64; 24];
        if let Some((x, y)) = p.into_affine().xy() {
            out[..6].copy_from_slice(&x.c0.0 .0);
            out[6..12].copy_from_slice(&x.c1.0 .0);
            out[12..18].copy_from_slice(&y.c0.0 .0);
            out[18..].copy_from_slice(&y.c1.0 .0);
        }
        out
    }
    fn g2_from_limbs(l: &[u64]) -> Self::G2 {
        use ark_bls12_381::{Fq, Fq2, G2Affine};
        if l.iter().all(|&w| w == 0) { return Self::G2::default(); }
        let fq = |w: &[u64]| Fq::new_unchecked(ark_ff::BigInt::new(w.try_into().unwrap()));
        G2Affine::new_unchecked(Fq2::new(fq(&l[..6]), fq(&l[6..12])), Fq2::new(fq(&l[12..18]), fq(&l[18..]))).into()
    }
}

fn as_limbs<F: ZkField>(v: &[F]) -> *const u64 {
    assert_eq!(std::mem::size_of::<F>(), 8 * F::LIMBS, "unexpected Fp layout");
    v.as_ptr() as *const u64
}
fn el<F: ZkField>(x: &F) -> *const u64 { x as *const F as *const u64 }
fn el_mut<F: ZkField>(x: &mut F) -> *mut u64 { x as *mut F as *mut u64 }
/// status -> the reference's behaviour: precondition codes panic with the reference's message text
fn check(rc: c_int) {
    if rc == 0 { return; }
    let msg = unsafe { std::ffi::CStr::from_ptr(ffi::zk_status_message(rc)) }.to_string_lossy().into_owned();
    panic!("{msg}");            // e.g. "Evaluated values must be a power of 2" (evaluation_form.rs:13)
}

/// HBM-resident table handle (RAII).  Not a reference item: what a prover that keeps its tables on the device holds.
pub struct DeviceTable<F: ZkField> { h: *mut ffi::zk_table, _f: std::marker::PhantomData<F> }
impl<F: ZkField> DeviceTable<F> {
    pub fn upload(v: &[F]) -> Self {
        let mut h = std::ptr::null_mut();
        check(unsafe { ffi::zk_table_upload(F::ID, as_limbs(v), v.len(), &mut h) });
        Self { h, _f: Default::default() }
    }
    pub fn alloc(len: usize) -> Self {
        let mut h = std::ptr::null_mut();
        check(unsafe { ffi::zk_table_alloc(F::ID, len, &mut h) });
        Self { h, _f: Default::default() }
    }
    fn adopt(h: *mut ffi::zk_table) -> Self { Self { h, _f: Default::default() } }
    pub fn len(&self) -> usize { unsafe { ffi::zk_table_len(self.h) } }
    pub fn fold(&self, var: usize, value: F) -> Self {
        let out = Self::alloc((self.len() / 2).max(1));
        check(unsafe { ffi::zk_mle_fold(self.h, var, el(&value), out.h, std::ptr::null_mut()) });
        out
    }
    pub fn download(&self) -> Vec<F> {
        let mut v = vec![F::zero(); self.len()];
        check(unsafe { ffi::zk_table_download(self.h, v.as_mut_ptr() as *mut u64) });
        v
    }
}
impl<F: ZkField> Drop for DeviceTable<F> { fn drop(&mut self) { unsafe { ffi::zk_table_free(self.h); } } }

/// every table of a SumPolynomial resident in HBM, product-major ([p * nfac + f]), as the library's array-of-handles argument
struct DeviceSum<F: ZkField> { tabs: Vec<DeviceTable<F>>, ptrs: Vec<*const ffi::zk_table>, nprod: usize, nfac: usize }
impl<F: ZkField> DeviceSum<F> {
    fn upload(products: &[Vec<&polynomials::multilinear::evaluation_form::MultilinearPolynomial<F>>]) -> Self {
        let (nprod, nfac) = (products.len(), products[0].len());
        let tabs: Vec<_> = products.iter().flat_map(|p| p.iter().map(|t| DeviceTable::<F>::upload(&t.evaluated_values))).collect();
        let ptrs = tabs.iter().map(|t| t.h as *const ffi::zk_table).collect();
        Self { tabs, ptrs, nprod, nfac }
    }
}

// =====================================================================================================================
// The reference's items, module for module (crate `polynomials` -> `polynomials::`, ...): same item names, public fields,
// argument names, argument and return types (tests/test_rust_shim_surface.py compares this file with the listing
// tests/golden/reference_api_surface.json).  `F: PrimeField` narrows to `F: ZkField`, `P: Pairing` to `P: ZkPairing`.
// =====================================================================================================================
pub mod polynomials {
    pub mod multilinear {
        pub mod evaluation_form {
            use crate::*;
            /// evaluation_form.rs:7-9
            #[derive(Debug, Clone, PartialEq)]
            pub struct MultilinearPolynomial<F: ZkField> { pub evaluated_values: Vec<F> }

            impl<F: ZkField> MultilinearPolynomial<F> {
                pub fn new(evaluated_values: &[F]) -> Self {                               // :12-18
                    assert!(evaluated_values.len().is_power_of_two(), "Evaluated values must be a power of 2");
                    Self { evaluated_values: evaluated_values.to_vec() }
                }
                pub fn evaluate(&self, values: &[F]) -> F {                                // :21-33
                    let t = DeviceTable::<F>::upload(&self.evaluated_values);
                    let mut out = F::zero();
                    check(unsafe { ffi::zk_mle_evaluate(t.h, as_limbs(values), values.len(), el_mut(&mut out)) });
                    out
                }
                pub fn convert_to_bytes(&self) -> Vec<u8> {                                // :35-43
                    let t = DeviceTable::<F>::upload(&self.evaluated_values);
                    let mut out = vec![0u8; 8 * F::LIMBS * self.evaluated_values.len()];
                    check(unsafe { ffi::zk_mle_to_bytes(t.h, out.as_mut_ptr()) });
                    out
                }
                pub fn number_of_variables(&self) -> u32 { self.evaluated_values.len().ilog2() }   // :45-47
                pub fn scalar_mul(&self, scalar: F) -> Self {                              // :49-57
                    let t = DeviceTable::<F>::upload(&self.evaluated_values);
                    let out = DeviceTable::<F>::alloc(t.len());
                    check(unsafe { ffi::zk_mle_scalar_mul(t.h, el(&scalar), out.h, std::ptr::null_mut()) });
                    Self { evaluated_values: out.download() }
                }
                /// partial_evaluate (:61-106): upload, one fold kernel, download.  Provers keep tables resident (`DeviceTable`)
                /// instead of paying PCIe per call.
                pub fn partial_evaluate(polynomial: &Vec<F>, evaluating_variable: usize, value: F) -> Self {
                    let t = DeviceTable::<F>::upload(polynomial);
                    Self { evaluated_values: t.fold(evaluating_variable, value).download() }
                }
                pub fn polynomial_tensor_add(w_b: &MultilinearPolynomial<F>, w_c: &MultilinearPolynomial<F>) -> MultilinearPolynomial<F> {   // :108-123
                    Self::tensor(w_b, w_c, false)
                }
                pub fn polynomial_tensor_mul(w_b: &MultilinearPolynomial<F>, w_c: &MultilinearPolynomial<F>) -> MultilinearPolynomial<F> {   // :125-143
                    Self::tensor(w_b, w_c, true)
                }
                pub fn add_polynomials(poly1: &MultilinearPolynomial<F>, poly2: &MultilinearPolynomial<F>) -> Self {                         // :145-163
                    let (a, b) = (DeviceTable::<F>::upload(&poly1.evaluated_values), DeviceTable::<F>::upload(&poly2.evaluated_values));
                    let out = DeviceTable::<F>::alloc(a.len());
                    check(unsafe { ffi::zk_mle_add(a.h, b.h, out.h, std::ptr::null_mut()) });          // length mismatch -> the reference's panic text
                    Self { evaluated_values: out.download() }
                }
                fn tensor(w_b: &MultilinearPolynomial<F>, w_c: &MultilinearPolynomial<F>, mul: bool) -> MultilinearPolynomial<F> {
                    let (b, c) = (DeviceTable::<F>::upload(&w_b.evaluated_values), DeviceTable::<F>::upload(&w_c.evaluated_values));
                    let out = DeviceTable::<F>::alloc(b.len() * c.len());
                    check(unsafe { if mul { ffi::zk_mle_tensor_mul(b.h, c.h, out.h, std::ptr::null_mut()) }
                                   else { ffi::zk_mle_tensor_add(b.h, c.h, out.h, std::ptr::null_mut()) } });
                    Self { evaluated_values: out.download() }
                }
            }
        }
    }
    pub mod composed {
        pub mod product_polynomial {
            use crate::polynomials::multilinear::evaluation_form::MultilinearPolynomial;
            use crate::*;
            /// product_polynomial.rs:6-8
            #[derive(Clone, Debug, PartialEq)]
            pub struct ProductPolynomial<F: ZkField> { pub polynomials: Vec<MultilinearPolynomial<F>> }
            impl<F: ZkField> ProductPolynomial<F> {
                pub fn new(polynomials: Vec<MultilinearPolynomial<F>>) -> Self {           // :11-24
                    let n = polynomials[0].number_of_variables();
                    assert!(polynomials.iter().all(|p| p.number_of_variables() == n), "different number of variables");
                    Self { polynomials }
                }
                pub fn evaluate(&self, values: &Vec<F>) -> F {                             // :26-34
                    self.polynomials.iter().fold(F::one(), |acc, p| acc * p.evaluate(values))
                }
                pub fn partial_evaluate(&self, evaluating_variable: usize, value: F) -> Vec<MultilinearPolynomial<F>> {   // :36-54
                    self.polynomials.iter().map(|p| MultilinearPolynomial::partial_evaluate(&p.evaluated_values, evaluating_variable, value)).collect()
                }
                pub fn multiply_polynomials_element_wise(&self) -> MultilinearPolynomial<F> {   // :58-73
                    assert!(self.polynomials.len() > 1, "more than one polynomial required for mul operation");
                    let d = DeviceSum::<F>::upload(&[self.polynomials.iter().collect::<Vec<_>>()]);
                    let out = DeviceTable::<F>::alloc(d.tabs[0].len());
                    check(unsafe { ffi::zk_prodpoly_reduce(d.ptrs.as_ptr(), d.nfac, out.h) });
                    MultilinearPolynomial { evaluated_values: out.download() }
                }
                pub fn convert_to_bytes(&self) -> Vec<u8> {                                // :75-83
                    self.polynomials.iter().flat_map(|p| p.convert_to_bytes()).collect()
                }
                pub fn degree(&self) -> usize { self.polynomials.len() }                   // :85-87
            }
        }
        pub mod sum_polynomial {
            use crate::polynomials::composed::product_polynomial::ProductPolynomial;
            use crate::polynomials::multilinear::evaluation_form::MultilinearPolynomial;
            use crate::*;
            /// sum_polynomial.rs:7-9
            #[derive(Clone, Debug, PartialEq)]
            pub struct SumPolynomial<F: ZkField> { pub product_polynomials: Vec<ProductPolynomial<F>> }
            impl<F: ZkField> SumPolynomial<F> {
                pub fn new(product_polynomials: Vec<ProductPolynomial<F>>) -> Self {       // :12-28
                    let n = product_polynomials[0].polynomials[0].number_of_variables();
                    assert!(product_polynomials.iter().all(|pp| pp.polynomials.iter().all(|p| p.number_of_variables() == n)),
                            "different number of variables");
                    Self { product_polynomials }
                }
                pub(crate) fn table_refs(&self) -> Vec<Vec<&MultilinearPolynomial<F>>> {
                    self.product_polynomials.iter().map(|pp| pp.polynomials.iter().collect()).collect()
                }
                pub fn evaluate(&self, values: &Vec<F>) -> F {                             // :30-38
                    let d = DeviceSum::<F>::upload(&self.table_refs());
                    let mut out = F::zero();
                    check(unsafe { ffi::zk_sumpoly_evaluate(d.ptrs.as_ptr(), d.nprod, d.nfac, as_limbs(values), values.len(), el_mut(&mut out)) });
                    out
                }
                pub fn partial_evaluate(&self, evaluating_variable: usize, value: F) -> Self {   // :40-53
                    Self { product_polynomials: self.product_polynomials.iter()
                        .map(|pp| ProductPolynomial { polynomials: pp.partial_evaluate(evaluating_variable, value) }).collect() }
                }
                pub fn add_polynomials_element_wise(&self) -> MultilinearPolynomial<F> {   // :57-76
                    assert!(self.product_polynomials.len() > 1, "more than one product polynomial required for add operation");
                    let d = DeviceSum::<F>::upload(&self.table_refs());
                    let out = DeviceTable::<F>::alloc(d.tabs[0].len());
                    check(unsafe { ffi::zk_sumpoly_reduce(d.ptrs.as_ptr(), d.nprod, d.nfac, out.h) });
                    MultilinearPolynomial { evaluated_values: out.download() }
                }
                pub fn convert_to_bytes(&self) -> Vec<u8> {                                // :78-86
                    self.product_polynomials.iter().flat_map(|pp| pp.convert_to_bytes()).collect()
                }
                pub fn degree(&self) -> usize { self.product_polynomials[0].degree() }     // :88-90
                pub fn number_of_variables(&self) -> u32 { self.product_polynomials[0].polynomials[0].number_of_variables() }   // :92-94
            }
        }
    }
    pub mod univariate {
        pub mod dense_univariate {
            use crate::*;
            /// dense_univariate.rs:4-6.  O(degree) host arithmetic: the round polynomials of a sumcheck have three coefficients.
            #[derive(Clone, Debug)]
            pub struct DenseUnivariatePolynomial<F: ZkField> { pub coefficients: Vec<F> }
            impl<F: ZkField> DenseUnivariatePolynomial<F> {
                pub fn new(coefficients: &Vec<F>) -> Self { Self { coefficients: coefficients.to_vec() } }   // :9-13
                pub fn degree(&self) -> u32 { self.coefficients.len() as u32 - 1 }          // :15-17
                pub fn evaluate_(&self, value: F) -> F { self.evaluate(value) }             // :23-35, the same value by another loop
                pub fn evaluate_advanced(&self, value: F) -> F { self.evaluate(value) }     // :38-46
                pub fn evaluate(&self, value: F) -> F {                                     // :57-68
                    let mut out = F::zero();
                    check(unsafe { ffi::zk_uni_evaluate(F::ID, as_limbs(&self.coefficients), self.coefficients.len(), el(&value), el_mut(&mut out)) });
                    out
                }
                pub fn lagrange_interpolate(x_values: &Vec<F>, y_values: &Vec<F>) -> DenseUnivariatePolynomial<F> {   // :74-98
                    let mut out = vec![F::zero(); x_values.len()];
                    check(unsafe { ffi::zk_uni_lagrange_interpolate(F::ID, as_limbs(x_values), as_limbs(y_values), x_values.len(), out.as_mut_ptr() as *mut u64) });
                    DenseUnivariatePolynomial { coefficients: out }
                }
            }
            pub fn multiply_polynomials<F: ZkField>(left: Vec<F>, right: Vec<F>) -> Vec<F> {   // :142-162
                let mut out = vec![F::zero(); left.len() + right.len() - 1];
                for (i, a) in left.iter().enumerate() { for (j, b) in right.iter().enumerate() { out[i + j] += *a * *b; } }
                out
            }
            pub fn add_polynomials<F: ZkField>(left: Vec<F>, right: Vec<F>) -> Vec<F> {       // :164-182
                let (long, short) = if left.len() > right.len() { (left, right) } else { (right, left) };
                long.iter().enumerate().map(|(k, c)| if k < short.len() { *c + short[k] } else { *c }).collect()
            }
        }
    }
}

pub mod transcripts {
    pub mod fiat_shamir {
        pub mod interface {
            use crate::ZkField;
            /// interface.rs:3-8
            pub trait FiatShamirTranscriptInterface {
                fn new() -> Self;
                fn append(&mut self, incoming_data: &[u8]);
                fn sample_random_challenge(&mut self) -> [u8; 32];
                fn random_challenge_as_field_element<F: ZkField>(&mut self) -> F;
            }
        }
        pub mod fiat_shamir_transcript {
            use super::interface::FiatShamirTranscriptInterface;
            use crate::*;
            /// fiat_shamir_transcript.rs:5-7: the sponge lives in the library (host Keccak-256 for the big absorbs, handed to the
            /// device for the rounds of a sumcheck)
            pub struct Transcript { pub(crate) h: *mut ffi::zk_transcript }
            impl FiatShamirTranscriptInterface for Transcript {
                fn new() -> Self {                                                          // :12-16
                    let mut h = std::ptr::null_mut();
                    check(unsafe { ffi::zk_transcript_new(&mut h) });
                    Self { h }
                }
                fn append(&mut self, incoming_data: &[u8]) {                                // :22-24
                    check(unsafe { ffi::zk_transcript_append(self.h, incoming_data.as_ptr(), incoming_data.len()) });
                }
                fn sample_random_challenge(&mut self) -> [u8; 32] {                         // :29-36
                    let mut out = [0u8; 32];
                    check(unsafe { ffi::zk_transcript_sample(self.h, out.as_mut_ptr()) });
                    out
                }
                fn random_challenge_as_field_element<F: ZkField>(&mut self) -> F {          // :38-43
                    let mut out = F::zero();
                    check(unsafe { ffi::zk_transcript_challenge(self.h, F::ID, el_mut(&mut out)) });
                    out
                }
            }
            impl Drop for Transcript { fn drop(&mut self) { unsafe { ffi::zk_transcript_free(self.h); } } }
        }
    }
}

pub mod sumcheck_protocol {
    pub mod basic_sumcheck {
        pub mod prover {
            use crate::polynomials::multilinear::evaluation_form::MultilinearPolynomial;
            use crate::transcripts::fiat_shamir::{fiat_shamir_transcript::Transcript, interface::FiatShamirTranscriptInterface};
            use crate::*;
            /// prover.rs:7-13
            pub struct Prover<F: ZkField> {
                pub initial_polynomial: MultilinearPolynomial<F>,
                pub initial_claimed_sum: F,
                pub transcript: Transcript,
                pub round_univariate_polynomials: Vec<MultilinearPolynomial<F>>,
                pub is_initialized: bool,
            }
            /// prover.rs:15-19
            pub struct SumcheckProof<F: ZkField> {
                pub initial_polynomial: MultilinearPolynomial<F>,
                pub initial_claimed_sum: F,
                pub round_univariate_polynomials: Vec<MultilinearPolynomial<F>>,
            }
            impl<F: ZkField> Prover<F> {
                pub fn init(polynomial_evaluated_values: &Vec<F>) -> Self {                 // :22-33
                    Prover {
                        initial_polynomial: MultilinearPolynomial::new(polynomial_evaluated_values),
                        initial_claimed_sum: polynomial_evaluated_values.iter().sum(),      // :28
                        transcript: Transcript::new(),
                        round_univariate_polynomials: Vec::new(),
                        is_initialized: true,
                    }
                }
                /// prove (:35-71): ONE library call on `self.transcript` (zk_sumcheck_basic_prove_on): the table absorb, every round's
                /// two half sums, challenge and fold; the transcript is left in the state the reference's prover leaves its own in.
                pub fn prove(&mut self) -> SumcheckProof<F> {
                    assert!(self.is_initialized, "Can't prove without init");
                    let n = self.initial_polynomial.number_of_variables() as usize;
                    let t = DeviceTable::<F>::upload(&self.initial_polynomial.evaluated_values);
                    let (mut claimed, mut rounds) = (F::zero(), vec![F::zero(); 2 * n.max(1)]);
                    check(unsafe { ffi::zk_sumcheck_basic_prove_on(t.h, self.transcript.h, el_mut(&mut claimed), rounds.as_mut_ptr() as *mut u64, std::ptr::null_mut()) });
                    self.round_univariate_polynomials = rounds.chunks(2).take(n).map(MultilinearPolynomial::new).collect();   // :56
                    SumcheckProof {
                        initial_polynomial: self.initial_polynomial.clone(),
                        initial_claimed_sum: self.initial_claimed_sum,
                        round_univariate_polynomials: self.round_univariate_polynomials.clone(),
                    }
                }
            }
            pub fn split_polynomial_and_sum_each<F: ZkField>(polynomial_evaluated_values: &Vec<F>) -> Vec<F> {   // :74-89
                let t = DeviceTable::<F>::upload(polynomial_evaluated_values);
                let mut out = vec![F::zero(); 2];
                check(unsafe { ffi::zk_mle_half_sums(t.h, out.as_mut_ptr() as *mut u64) });
                out
            }
            pub fn field_element_to_bytes<F: ZkField>(field_element: F) -> Vec<u8> {          // :91-93
                let mut out = vec![0u8; 8 * F::LIMBS];
                check(unsafe { ffi::zk_fe_to_bytes_be(F::ID, el(&field_element), out.as_mut_ptr()) });
                out
            }
        }
        pub mod verifier {
            use super::prover::SumcheckProof;
            use crate::transcripts::fiat_shamir::{fiat_shamir_transcript::Transcript, interface::FiatShamirTranscriptInterface};
            use crate::*;
            /// verifier.rs:8-12
            pub struct Verifier<F: ZkField> { pub transcript: Transcript, pub is_initialized: bool, _phantom: std::marker::PhantomData<F> }
            impl<F: ZkField> Verifier<F> {
                pub fn init() -> Self { Self { transcript: Transcript::new(), is_initialized: true, _phantom: Default::default() } }   // :15-21
                /// verify (:23-71): the claim chain on the host, the final `evaluate` as GPU folds
                pub fn verify(&mut self, proof: SumcheckProof<F>) -> bool {
                    assert!(self.is_initialized, "Can't verify without init");
                    let t = DeviceTable::<F>::upload(&proof.initial_polynomial.evaluated_values);
                    let flat: Vec<F> = proof.round_univariate_polynomials.iter().flat_map(|p| p.evaluated_values.iter().cloned()).collect();
                    let mut ok: c_int = 0;
                    check(unsafe { ffi::zk_sumcheck_basic_verify(t.h, el(&proof.initial_claimed_sum), as_limbs(&flat), proof.round_univariate_polynomials.len(), &mut ok) });
                    ok == 1
                }
            }
        }
    }
    pub mod gkr_sumcheck {
        /// sumcheck_gkr_protocol.rs:8-21
        pub mod sumcheck_gkr_protocol {
            use crate::polynomials::composed::sum_polynomial::SumPolynomial;
            use crate::polynomials::univariate::dense_univariate::DenseUnivariatePolynomial;
            use crate::transcripts::fiat_shamir::fiat_shamir_transcript::Transcript;
            use crate::*;
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
                check(unsafe { ffi::zk_sumcheck_gkr_prove(d.ptrs.as_ptr(), d.nprod, d.nfac, el(&claimed_sum), transcript.h,
                                                          coeffs.as_mut_ptr() as *mut u64, chal.as_mut_ptr() as *mut u64) });
                SumcheckProverProof {
                    claimed_sum,
                    round_univariate_polynomials: coeffs.chunks(ncoef).take(n).map(|c| DenseUnivariatePolynomial::new(&c.to_vec())).collect(),
                    random_challenges: chal[..n].to_vec(),
                }
            }
            /// verify (:69-111)
            pub fn verify<F: ZkField>(proof: &SumcheckProverProof<F>, transcript: &mut Transcript) -> SumcheckVerifierProof<F> {
                let n = proof.round_univariate_polynomials.len();
                let ncoef = proof.round_univariate_polynomials.first().map_or(0, |p| p.coefficients.len());
                let flat: Vec<F> = proof.round_univariate_polynomials.iter().flat_map(|p| p.coefficients.iter().cloned()).collect();
                let (mut chal, mut last, mut ok) = (vec![F::zero(); n.max(1)], F::zero(), 0 as c_int);
                check(unsafe { ffi::zk_sumcheck_gkr_verify(F::ID, el(&proof.claimed_sum), as_limbs(&flat), n, ncoef, transcript.h,
                                                           chal.as_mut_ptr() as *mut u64, el_mut(&mut last), &mut ok) });
                SumcheckVerifierProof { is_proof_valid: ok == 1, random_challenges: chal[..n].to_vec(), last_claimed_sum: last }
            }
            /// generate_round_univariate (:113-143): the evaluations at 0..=degree of the current round
            pub fn generate_round_univariate<F: ZkField>(current_polynomial: &SumPolynomial<F>) -> Vec<F> {
                let d = DeviceSum::<F>::upload(&current_polynomial.table_refs());
                let mut out = vec![F::zero(); current_polynomial.degree() + 1];
                check(unsafe { ffi::zk_sumpoly_round_evals(d.ptrs.as_ptr(), d.nprod, d.nfac, out.as_mut_ptr() as *mut u64) });
                out
            }
            pub fn univariate_to_bytes<F: ZkField>(univariate_poly: &[F]) -> Vec<u8> {       // :145-150: canonical integers, little-endian
                let mut canon = vec![0u64; F::LIMBS * univariate_poly.len()];
                check(unsafe { ffi::zk_vec_to_canonical(F::ID, as_limbs(univariate_poly), univariate_poly.len(), canon.as_mut_ptr()) });
                canon.iter().flat_map(|w| w.to_le_bytes()).collect()
            }
            pub fn field_element_to_bytes<F: ZkField>(field_element: F) -> Vec<u8> {        // :152-154
                crate::sumcheck_protocol::basic_sumcheck::prover::field_element_to_bytes(field_element)
            }
        }
    }
}

pub mod circuit {
    /// arithmetic_circuit.rs:5-30
    pub mod arithmetic_circuit {
        use crate::polynomials::multilinear::evaluation_form::MultilinearPolynomial;
        use crate::*;
        pub enum Operator { Add, Mul }
        pub struct Gate { pub left_index: usize, pub right_index: usize, pub output_index: usize, pub operator: Operator }
        pub struct Layer { pub gates: Vec<Gate> }
        pub struct Circuit<F: ZkField> { pub layers: Vec<Layer>, _phantom: std::marker::PhantomData<F> }
        pub struct CircuitEvaluationResult<F: ZkField> { pub output: Vec<F>, pub layer_evaluations: Vec<Vec<F>> }
        impl Gate {
            pub fn new(left_index: usize, right_index: usize, output_index: usize, operator: Operator) -> Self {   // :34-47
                Self { left_index, right_index, output_index, operator }
            }
        }
        impl Layer { pub fn new(gates: Vec<Gate>) -> Self { Self { gates } } }             // :51-53
        impl<F: ZkField> Circuit<F> {
            pub fn new(layers: Vec<Layer>) -> Self { Self { layers, _phantom: Default::default() } }   // :58-63
            /// the library's flat form: all gates, layer 0 (the output layer) first, and the per-layer counts
            pub(crate) fn flat(&self) -> (Vec<ffi::zk_gate>, Vec<usize>) {
                let gates = self.layers.iter().flat_map(|l| l.gates.iter().map(|g| ffi::zk_gate {
                    left: g.left_index as u64, right: g.right_index as u64, out: g.output_index as u64,
                    op: match g.operator { Operator::Add => 0, Operator::Mul => 1 } })).collect();
                (gates, self.layers.iter().map(|l| l.gates.len()).collect())
            }
            pub fn evaluate(&mut self, values: Vec<F>) -> CircuitEvaluationResult<F> {     // :65-109
                let (gates, counts) = self.flat();
                let nl = counts.len();
                let total = unsafe { ffi::zk_circuit_eval_size(gates.as_ptr(), counts.as_ptr(), nl, values.len()) };
                let (mut sizes, mut evals) = (vec![0usize; nl + 1], vec![F::zero(); total]);
                check(unsafe { ffi::zk_circuit_evaluate(F::ID, gates.as_ptr(), counts.as_ptr(), nl, as_limbs(&values), values.len(),
                                                        sizes.as_mut_ptr(), evals.as_mut_ptr() as *mut u64) });
                let mut off = 0;
                let layer_evaluations: Vec<Vec<F>> = sizes.iter().map(|&s| { let v = evals[off..off + s].to_vec(); off += s; v }).collect();
                CircuitEvaluationResult { output: layer_evaluations[0].clone(), layer_evaluations }
            }
            pub fn w_i_polynomial(circuit_evaluation: &CircuitEvaluationResult<F>, layer_index: usize) -> MultilinearPolynomial<F> {   // :114-124
                assert!(layer_index < circuit_evaluation.layer_evaluations.len(), "layer index out of bounds");
                MultilinearPolynomial::new(&circuit_evaluation.layer_evaluations[layer_index])
            }
            pub fn add_i_and_mul_i_mle(&mut self, layer_index: usize) -> (MultilinearPolynomial<F>, MultilinearPolynomial<F>) {        // :126-163
                let gates: Vec<ffi::zk_gate> = self.layers[layer_index].gates.iter().map(|g| ffi::zk_gate {
                    left: g.left_index as u64, right: g.right_index as u64, out: g.output_index as u64,
                    op: match g.operator { Operator::Add => 0, Operator::Mul => 1 } }).collect();
                let (mut a, mut m) = (std::ptr::null_mut(), std::ptr::null_mut());
                check(unsafe { ffi::zk_circuit_add_mul_mle(F::ID, gates.as_ptr(), gates.len(), layer_index, &mut a, &mut m) });
                let (a, m) = (DeviceTable::<F>::adopt(a), DeviceTable::<F>::adopt(m));
                (MultilinearPolynomial { evaluated_values: a.download() }, MultilinearPolynomial { evaluated_values: m.download() })
            }
        }
        pub fn num_of_layer_variables(layer_index: usize) -> usize { unsafe { ffi::zk_num_of_layer_variables(layer_index) } }   // :166-178
        pub fn convert_to_binary_and_to_decimal(layer_index: usize, variable_a: usize, variable_b: usize, variable_c: usize) -> usize {   // :180-196
            unsafe { ffi::zk_wiring_index(layer_index, variable_a, variable_b, variable_c) }
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
    fn sumcheck_proofs(&self, nlayers: usize) -> Vec<sumcheck_protocol::gkr_sumcheck::sumcheck_gkr_protocol::SumcheckProverProof<F>> {
        use polynomials::univariate::dense_univariate::DenseUnivariatePolynomial;
        let mut off = 0;
        Self::rounds(nlayers).iter().enumerate().map(|(l, &r)| {
            let p = sumcheck_protocol::gkr_sumcheck::sumcheck_gkr_protocol::SumcheckProverProof {
                claimed_sum: self.claims[l],
                round_univariate_polynomials: self.coeffs[3 * off..3 * (off + r)].chunks(3).map(|c| DenseUnivariatePolynomial::new(&c.to_vec())).collect(),
                random_challenges: self.chal[off..off + r].to_vec(),
            };
            off += r;
            p
        }).collect()
    }
    fn from_proofs(proofs: &[sumcheck_protocol::gkr_sumcheck::sumcheck_gkr_protocol::SumcheckProverProof<F>]) -> (Vec<F>, Vec<F>) {
        (proofs.iter().map(|p| p.claimed_sum).collect(),
         proofs.iter().flat_map(|p| p.round_univariate_polynomials.iter().flat_map(|u| u.coefficients.iter().cloned())).collect())
    }
}

pub mod gkr {
    /// gkr_protocol.rs:17-23, :26, :146
    pub mod gkr_protocol {
        use crate::circuit::arithmetic_circuit::Circuit;
        use crate::sumcheck_protocol::gkr_sumcheck::sumcheck_gkr_protocol::SumcheckProverProof;
        use crate::*;
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
                                             b.out.as_mut_ptr() as *mut u64, &mut b.out_len, el_mut(&mut b.claimed),
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
    /// utils.rs:8-82: the pieces gkr_protocol::prove is made of, for callers that assemble their own layer (the library's prove runs
    /// them fused on the device; these are the same operations one call at a time)
    pub mod utils {
        use crate::polynomials::composed::{product_polynomial::ProductPolynomial, sum_polynomial::SumPolynomial};
        use crate::polynomials::multilinear::evaluation_form::MultilinearPolynomial;
        use crate::*;
        pub fn compute_fbc_polynomial<F: ZkField>(add_i_bc: MultilinearPolynomial<F>, mul_i_bc: MultilinearPolynomial<F>,
                                                  w_b_polynomial: &MultilinearPolynomial<F>, w_c_polynomial: &MultilinearPolynomial<F>) -> SumPolynomial<F> {   // :8-21
            let add_wb_wc = MultilinearPolynomial::polynomial_tensor_add(w_b_polynomial, w_c_polynomial);
            let mul_wb_wc = MultilinearPolynomial::polynomial_tensor_mul(w_b_polynomial, w_c_polynomial);
            SumPolynomial::new(vec![ProductPolynomial::new(vec![add_i_bc, add_wb_wc]), ProductPolynomial::new(vec![mul_i_bc, mul_wb_wc])])
        }
        pub fn compute_new_add_i_mul_i<F: ZkField>(alpha: F, beta: F, add_i_abc: MultilinearPolynomial<F>, mul_i_abc: MultilinearPolynomial<F>,
                                                   rb_values: &[F], rc_values: &[F]) -> (MultilinearPolynomial<F>, MultilinearPolynomial<F>) {   // :23-68
            let chain = |t: &MultilinearPolynomial<F>, r: &[F]| {
                let mut d = DeviceTable::<F>::upload(&t.evaluated_values);
                for v in r { d = d.fold(0, *v); }
                MultilinearPolynomial { evaluated_values: d.download() }
            };
            let combine = |t: &MultilinearPolynomial<F>| MultilinearPolynomial::add_polynomials(&chain(t, rb_values).scalar_mul(alpha), &chain(t, rc_values).scalar_mul(beta));
            (combine(&add_i_abc), combine(&mul_i_abc))
        }
        pub fn evaluate_wb_wc<F: ZkField>(wb_poly: &MultilinearPolynomial<F>, wc_poly: &MultilinearPolynomial<F>, sumcheck_challenges: &[F]) -> (F, F) {   // :70-82
            let middle = sumcheck_challenges.len() / 2;
            (wb_poly.evaluate(&sumcheck_challenges[..middle]), wc_poly.evaluate(&sumcheck_challenges[middle..]))
        }
    }
    /// succinct_gkr_protocol.rs:23-32, :35, :172
    pub mod succinct_gkr_protocol {
        use crate::circuit::arithmetic_circuit::Circuit;
        use crate::multilinear_kzg::multilinear_kzg::MultilinearKZGProof;
        use crate::multilinear_kzg::trusted_setup::TrustedSetup;
        use crate::sumcheck_protocol::gkr_sumcheck::sumcheck_gkr_protocol::SumcheckProverProof;
        use crate::*;
        #[derive(Clone)]
        pub struct SuccinctProof<F: ZkField, P: ZkPairing> {
            pub circuit_output: Vec<F>,
            pub claimed_sum: F,
            pub sumcheck_proofs: Vec<SumcheckProverProof<F>>,
            pub wb_evaluations: Vec<F>,
            pub wc_evaluations: Vec<F>,
            pub input_polynomial_commitment: P::G1,
            pub input_rb_proof: MultilinearKZGProof<F, P>,
            pub input_rc_proof: MultilinearKZGProof<F, P>,
        }
        /// prove_succinct (:35-169): the GKR proof + commit(inputs) (:42-44) + open(inputs, rb), open(inputs, rc) (:154-157).
        /// The library's entry point is BLS12-381 Fr (F::ID 0), the reference's instantiation.
        pub fn prove_succinct<F: ZkField, P: ZkPairing>(circuit: &mut Circuit<F>, inputs: &[F], trusted_setup: &TrustedSetup<P>) -> SuccinctProof<F, P> {
            assert_eq!(F::ID, 0, "succinct GKR runs on BLS12-381 Fr");
            let (gates, counts) = circuit.flat();
            let nl = counts.len();
            let mut b = FlatGkr::<F>::alloc(nl);
            let (mut com, mut rb_ev, mut rc_ev) = ([0u64; 12], F::zero(), F::zero());
            let (mut rb, mut rc) = (vec![0u64; 12 * nl], vec![0u64; 12 * nl]);
            check(unsafe { ffi::zk_gkr_prove_succinct(gates.as_ptr(), counts.as_ptr(), nl, as_limbs(inputs), inputs.len(), trusted_setup.device().g1,
                                                      trusted_setup.g2_powers_of_tau.len(), b.out.as_mut_ptr() as *mut u64, &mut b.out_len,
                                                      el_mut(&mut b.claimed), b.claims.as_mut_ptr() as *mut u64,
                                                      b.coeffs.as_mut_ptr() as *mut u64, b.chal.as_mut_ptr() as *mut u64,
                                                      b.wb.as_mut_ptr() as *mut u64, b.wc.as_mut_ptr() as *mut u64, com.as_mut_ptr(),
                                                      el_mut(&mut rb_ev), rb.as_mut_ptr(), el_mut(&mut rc_ev), rc.as_mut_ptr()) });
            let pts = |v: &[u64]| v.chunks(12).map(P::g1_from_limbs).collect::<Vec<_>>();
            SuccinctProof { circuit_output: b.out[..b.out_len].to_vec(), claimed_sum: b.claimed, sumcheck_proofs: b.sumcheck_proofs(nl),
                            wb_evaluations: b.wb[..nl - 1].to_vec(), wc_evaluations: b.wc[..nl - 1].to_vec(),
                            input_polynomial_commitment: P::g1_from_limbs(&com),
                            input_rb_proof: MultilinearKZGProof { evaluation: rb_ev, proofs: pts(&rb) },
                            input_rc_proof: MultilinearKZGProof { evaluation: rc_ev, proofs: pts(&rc) } }
        }
        /// verify_succinct (:172-285): GKR rounds + two KZG verifications (n + 1 pairings each, host side of the library)
        pub fn verify_succinct<F: ZkField, P: ZkPairing>(circuit: &mut Circuit<F>, proof: SuccinctProof<F, P>, trusted_setup: &TrustedSetup<P>) -> bool {
            let (gates, counts) = circuit.flat();
            let (claims, coeffs) = FlatGkr::<F>::from_proofs(&proof.sumcheck_proofs);
            let flat = |p: &MultilinearKZGProof<F, P>| p.proofs.iter().flat_map(|q| P::g1_to_limbs(q)).collect::<Vec<u64>>();
            let (com, rb, rc) = (P::g1_to_limbs(&proof.input_polynomial_commitment), flat(&proof.input_rb_proof), flat(&proof.input_rc_proof));
            let g2: Vec<u64> = trusted_setup.g2_powers_of_tau.iter().flat_map(|q| P::g2_to_limbs(q)).collect();
            let mut ok: c_int = 0;
            check(unsafe { ffi::zk_gkr_verify_succinct(gates.as_ptr(), counts.as_ptr(), counts.len(), as_limbs(&proof.circuit_output),
                                                       proof.circuit_output.len(), as_limbs(&claims), as_limbs(&coeffs),
                                                       as_limbs(&proof.wb_evaluations), as_limbs(&proof.wc_evaluations), com.as_ptr(),
                                                       el(&proof.input_rb_proof.evaluation), rb.as_ptr(), proof.input_rb_proof.proofs.len(),
                                                       el(&proof.input_rc_proof.evaluation), rc.as_ptr(), proof.input_rc_proof.proofs.len(),
                                                       g2.as_ptr(), trusted_setup.g2_powers_of_tau.len(), &mut ok) });
            ok == 1
        }
    }
}

pub mod multilinear_kzg {
    pub mod trusted_setup {
        use crate::*;
        /// the setup's G1 powers resident in HBM: uploaded (or produced) once per TrustedSetup, then every commit / opening runs against them
        pub struct DeviceSetup { pub(crate) g1: *mut ffi::zk_g1_bases }
        impl Drop for DeviceSetup { fn drop(&mut self) { unsafe { ffi::zk_g1_bases_free(self.g1); } } }
        /// trusted_setup.rs:5-8.  The two public fields are the reference's; the device copy of the G1 powers is cached behind them.
        pub struct TrustedSetup<P: ZkPairing> {
            pub g1_powers_of_tau: Vec<P::G1>,
            pub g2_powers_of_tau: Vec<P::G2>,
            device: std::cell::OnceCell<DeviceSetup>,
        }
        impl<P: ZkPairing> TrustedSetup<P> {
            /// initialize_setup (:11-22): compute_lagrange_basis + 2^n fixed-base [L_i(tau)]G on the device, the n G2 powers on the host
            pub fn initialize_setup<F: ZkField>(taus: &[F]) -> Self {
                let mut g1 = std::ptr::null_mut();
                check(unsafe { ffi::zk_kzg_setup_g1(as_limbs(taus), taus.len(), &mut g1) });
                let n = unsafe { ffi::zk_g1_bases_len(g1) };
                let mut affine = vec![0u64; 12 * n];
                check(unsafe { ffi::zk_g1_bases_download(g1, affine.as_mut_ptr()) });
                let device = std::cell::OnceCell::new();
                let _ = device.set(DeviceSetup { g1 });
                Self { g1_powers_of_tau: affine.chunks(12).map(P::g1_from_limbs).collect(), g2_powers_of_tau: compute_g2_powers_of_tau::<P, F>(taus), device }
            }
            /// a setup assembled from its two public fields (the reference's struct literal); the G1 powers are uploaded at first use
            pub fn from_powers(g1_powers_of_tau: Vec<P::G1>, g2_powers_of_tau: Vec<P::G2>) -> Self {
                Self { g1_powers_of_tau, g2_powers_of_tau, device: std::cell::OnceCell::new() }
            }
            pub(crate) fn device(&self) -> &DeviceSetup {
                self.device.get_or_init(|| {
                    let affine: Vec<u64> = self.g1_powers_of_tau.iter().flat_map(|p| P::g1_to_limbs(p)).collect();
                    let mut g1 = std::ptr::null_mut();
                    check(unsafe { ffi::zk_g1_bases_upload(affine.as_ptr(), self.g1_powers_of_tau.len(), &mut g1) });
                    DeviceSetup { g1 }
                })
            }
            /// Optional, once per setup: window-shifted copies of the G1 powers (ceil(256 / 22) x 128 bytes per point of HBM), after which every
            /// commit_to_polynomial against this setup uses 22-bit windows on one bucket set.  The commitments are the same group elements.
            pub fn precompute_for_commits(&self) { check(unsafe { ffi::zk_g1_bases_precompute(self.device().g1, 0) }); }
        }
        pub fn compute_g2_powers_of_tau<P: ZkPairing, F: ZkField>(taus: &[F]) -> Vec<P::G2> {   // :62-72
            assert!(taus.len() > 0, "requires at least one variable");
            let mut g2 = vec![0u64; 24 * taus.len()];
            check(unsafe { ffi::zk_kzg_setup_g2(as_limbs(taus), taus.len(), g2.as_mut_ptr()) });
            g2.chunks(24).map(P::g2_from_limbs).collect()
        }
        pub fn generate_values_for_tau<F: ZkField>(number_of_variables: usize) -> Vec<F> {        // :76-87
            use ark_ff::UniformRand;
            let mut rng = rand::thread_rng();
            (0..number_of_variables).map(|_| F::rand(&mut rng)).collect()
        }
    }
    pub mod multilinear_kzg {
        use super::trusted_setup::TrustedSetup;
        use crate::polynomials::multilinear::evaluation_form::MultilinearPolynomial;
        use crate::*;
        /// multilinear_kzg.rs:10-14
        pub struct MultilinearKZG<F: ZkField, P: ZkPairing> { _phantom_f: std::marker::PhantomData<F>, _phantom_p: std::marker::PhantomData<P> }
        /// multilinear_kzg.rs:16-20
        #[derive(Clone, Debug)]
        pub struct MultilinearKZGProof<F: ZkField, P: ZkPairing> { pub evaluation: F, pub proofs: Vec<P::G1> }

        impl<F: ZkField, P: ZkPairing> MultilinearKZG<F, P> {
            /// commit_to_polynomial (:25-45): one Pippenger MSM over the resident G1 powers
            pub fn commit_to_polynomial(polynomial: &MultilinearPolynomial<F>, trusted_setup: &TrustedSetup<P>) -> P::G1 {
                assert_eq!(polynomial.evaluated_values.len(), trusted_setup.g1_powers_of_tau.len(), "Polynomial evaluation must match g1 length");
                let t = DeviceTable::<F>::upload(&polynomial.evaluated_values);
                let mut out = [0u64; 12];
                check(unsafe { ffi::zk_kzg_commit(t.h, trusted_setup.device().g1, out.as_mut_ptr()) });
                P::g1_from_limbs(&out)
            }
            /// open_and_prove (:50-126): evaluate + the n quotient MSMs of 2^(n-1) .. 1 terms on pre-summed bases
            pub fn open_and_prove(polynomial: &MultilinearPolynomial<F>, trusted_setup: &TrustedSetup<P>, opening_values: &[F]) -> MultilinearKZGProof<F, P> {
                let t = DeviceTable::<F>::upload(&polynomial.evaluated_values);
                let (mut ev, mut proofs) = (F::zero(), vec![0u64; 12 * opening_values.len().max(1)]);
                check(unsafe { ffi::zk_kzg_open(t.h, trusted_setup.device().g1, std::ptr::null(), as_limbs(opening_values), opening_values.len(),
                                                trusted_setup.g2_powers_of_tau.len(), el_mut(&mut ev), proofs.as_mut_ptr()) });
                MultilinearKZGProof { evaluation: ev, proofs: proofs.chunks(12).take(opening_values.len()).map(P::g1_from_limbs).collect() }
            }
            /// verify (:131-158): n + 1 pairings on the host side of the library
            pub fn verify(trusted_setup: &TrustedSetup<P>, commitment: &P::G1, opening_values: &[F], proof: &MultilinearKZGProof<F, P>) -> bool {
                let c = P::g1_to_limbs(commitment);
                let prs: Vec<u64> = proof.proofs.iter().flat_map(|p| P::g1_to_limbs(p)).collect();
                let g2: Vec<u64> = trusted_setup.g2_powers_of_tau.iter().flat_map(|q| P::g2_to_limbs(q)).collect();
                let mut ok: c_int = 0;
                check(unsafe { ffi::zk_kzg_verify(c.as_ptr(), as_limbs(opening_values), opening_values.len(), el(&proof.evaluation),
                                                  prs.as_ptr(), proof.proofs.len(), g2.as_ptr(), trusted_setup.g2_powers_of_tau.len(), &mut ok) });
                ok == 1
            }
        }
    }
}
