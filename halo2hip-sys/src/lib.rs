//! halo2hip-sys -- the `unsafe` half of the drop-in: halo2_proofs forbids unsafe code (`halo2_proofs/src/lib.rs:23`), so
//! the `extern "C"` block for libhalo2hip.so (include/halo2hip.h) and the reinterpretation of halo2curves' types as limb
//! arrays live here.  halo2_proofs gains two dispatch lines (patches/0001-arithmetic-dispatch.patch):
//!
//! ```ignore
//! if let Some(r) = halo2hip_sys::try_multiexp::<C>(coeffs, bases) { return r; }   // best_multiexp, arithmetic.rs:133
//! if halo2hip_sys::try_fft(a, &omega, log_n) { return; }                          // best_fft,      arithmetic.rs:184
//! ```
//!
//! Every wrapper returns `None` / `false` when the engine does not take the call (another curve, a size below the
//! threshold, no GPU, any non-zero status): the caller then runs the original CPU body.  Nothing panics or unwinds across
//! the FFI boundary.
//!
//! Not compiled in the repository that ships it (its build image has no Rust toolchain): `tests/test_binding.py` checks the
//! extern block against the header and the shared library, and that the patches apply to the reference tree.
#![allow(non_camel_case_types)]
#![allow(clippy::missing_safety_doc)]

pub mod evalh;
pub mod ffi;

use ff::{Field, PrimeField};
use halo2curves::bn256::{Fr, G1Affine, G1};
use halo2curves::CurveAffine;
use std::any::TypeId;
use std::sync::atomic::{AtomicU8, Ordering};

pub const H2HIP_OK: i32 = 0;
pub const H2HIP_EINVAL: i32 = 1;
pub const H2HIP_EDEVICE: i32 = 2;
pub const H2HIP_ENOMEM: i32 = 3;

/// Initialise the engine on the given GPUs (one context, stream and worker thread each; host-pointer MSMs shard over
/// them).  Optional: the first call initialises lazily on the current device, or on `HALO2_HIP_DEVICES`.
pub fn init(device_ids: &[i32]) -> Result<(), String> {
    let rc = unsafe { ffi::h2hip_init(device_ids.as_ptr(), device_ids.len() as i32) };
    if rc == 0 {
        Ok(())
    } else {
        Err(last_error())
    }
}

pub fn shutdown() {
    unsafe { ffi::h2hip_shutdown() }
}

pub fn last_error() -> String {
    unsafe { std::ffi::CStr::from_ptr(ffi::h2hip_last_error()) }.to_string_lossy().into_owned()
}

/// The layout the engine assumes, checked once per process: `Fr` = 4 x u64 Montgomery limbs, `G1Affine` = x || y with the
/// identity as (0, 0), `G1` = x || y || z.  halo2curves' types are not `#[repr(C)]`; `SerdeObject::to_raw_bytes` is the
/// layout-defined view ("uncompressed, internal Montgomery representation", halo2_proofs/src/helpers.rs:13-19), so the
/// in-memory bytes of a few values are compared with it.
fn layout_ok() -> bool {
    // 0 = not checked yet, 1 = ok, 2 = mismatch.  An atomic rather than `static mut` (a hard error under the 2024 edition's
    // static_mut_refs lint) or OnceLock (Rust 1.70; the reference's toolchain pin is older).  Two threads may both run the check
    // the first time; they store the same answer.
    static STATE: AtomicU8 = AtomicU8::new(0);
    match STATE.load(Ordering::Acquire) {
        1 => return true,
        2 => return false,
        _ => {}
    }
    use halo2curves::serde::SerdeObject;
    let mut ok = std::mem::size_of::<Fr>() == 32 && std::mem::size_of::<G1Affine>() == 64 && std::mem::size_of::<G1>() == 96;
    if ok {
        let raw_of = |p: *const u8, n: usize| unsafe { std::slice::from_raw_parts(p, n) }.to_vec();
        let two = Fr::one() + Fr::one();
        ok &= raw_of(&two as *const Fr as *const u8, 32) == two.to_raw_bytes();
        let g = G1Affine::generator();
        ok &= raw_of(&g as *const G1Affine as *const u8, 64) == g.to_raw_bytes();
        let id = G1Affine::default();
        ok &= raw_of(&id as *const G1Affine as *const u8, 64) == vec![0u8; 64];
    }
    STATE.store(if ok { 1 } else { 2 }, Ordering::Release);
    ok
}

fn is<A: 'static, B: 'static>() -> bool {
    TypeId::of::<A>() == TypeId::of::<B>()
}

/// `best_multiexp` on the GPU: `Some(sum)` when the engine took the call.
pub fn try_multiexp<C: CurveAffine>(coeffs: &[C::Scalar], bases: &[C]) -> Option<C::Curve> {
    if !is::<C, G1Affine>() || coeffs.len() != bases.len() || coeffs.len() < unsafe { ffi::h2hip_msm_min_n() } || !layout_ok() {
        return None;
    }
    let mut out = [0u64; 12];
    let rc = unsafe { ffi::h2hip_msm_bn254(coeffs.as_ptr() as *const u64, bases.as_ptr() as *const u64, coeffs.len(), out.as_mut_ptr()) };
    if rc != 0 || std::mem::size_of::<C::Curve>() != 96 {
        return None;
    }
    Some(unsafe { std::mem::transmute_copy::<[u64; 12], C::Curve>(&out) }) // C::Curve = G1 = (x, y, z)
}

/// `count` commitments over the same bases in one call (the advice-column loop of plonk/prover.rs:361-365).
pub fn try_multiexp_batch<C: CurveAffine>(columns: &[&[C::Scalar]], bases: &[C]) -> Option<Vec<C::Curve>> {
    if !is::<C, G1Affine>() || columns.is_empty() || !layout_ok() || std::mem::size_of::<C::Curve>() != 96 {
        return None;
    }
    let n = columns[0].len();
    if n > bases.len() || columns.iter().any(|c| c.len() != n) {
        return None;
    }
    let ptrs: Vec<*const u64> = columns.iter().map(|c| c.as_ptr() as *const u64).collect();
    let mut out = vec![[0u64; 12]; columns.len()];
    let rc = unsafe { ffi::h2hip_msm_bn254_batch(ptrs.as_ptr(), bases.as_ptr() as *const u64, n, columns.len(), out.as_mut_ptr() as *mut u64) };
    if rc != 0 {
        return None;
    }
    Some(out.iter().map(|o| unsafe { std::mem::transmute_copy::<[u64; 12], C::Curve>(o) }).collect())
}

/// `best_fft` on the GPU (G = bn256::Fr, or G = bn256::G1 for the curve-point FFT): true when the engine took the call (`a` then holds the transform).
/// Parity is defined for `omega` of exact order 2^log_n (every in-crate caller); anything else keeps the CPU body.
pub fn try_fft<G: 'static, S: 'static>(a: &mut [G], omega: &S, log_n: u32) -> bool {
    if log_n >= usize::BITS || a.len() != 1usize << log_n {
        return false;
    }
    if is::<G, G1>() {
        // best_fft over curve points (g_to_lagrange, arithmetic.rs:285): one 254-bit scalar multiplication per butterfly
        return match fft_guards::<Fr, S>(omega, log_n) {
            Some(w) => unsafe { ffi::h2hip_fft_bn254_g1(a.as_mut_ptr() as *mut u64, w as *const Fr as *const u64, log_n) == 0 },
            None => false,
        };
    }
    match fft_guards::<G, S>(omega, log_n) {
        Some(w) => unsafe { ffi::h2hip_ntt_bn254_fr(a.as_mut_ptr() as *mut u64, w as *const Fr as *const u64, log_n) == 0 },
        None => false,
    }
}

/// Keep `bases` (a `ParamsKZG`'s `g` or `g_lagrange`) on the GPU with its fixed-base window table until `unpin_bases`.
/// Keyed by the slice's address; a no-op for other curves.  Returns whether the engine holds the array now.
pub fn pin_bases<C: 'static>(bases: &[C]) -> bool {
    if !is::<C, G1Affine>() || bases.is_empty() || !layout_ok() {
        return false;
    }
    unsafe { ffi::h2hip_bases_pin(bases.as_ptr() as *const u64, bases.len()) == 0 }
}

/// Drop a pinned array (before its `Vec` is freed or rewritten).  Harmless when the array was never pinned.
pub fn unpin_bases<C: 'static>(bases: &[C]) {
    if is::<C, G1Affine>() && !bases.is_empty() {
        unsafe { ffi::h2hip_bases_unpin(bases.as_ptr() as *const std::os::raw::c_void) };
    }
}

/// `g_to_lagrange` (arithmetic.rs:277-301) for `ParamsKZG::downsize`: affine in, affine out.
pub fn try_g_to_lagrange<C: CurveAffine>(g: &[C], k: u32) -> Option<Vec<C>> {
    if !is::<C, G1Affine>() || g.len() != 1usize << k || !layout_ok() {
        return None;
    }
    let mut out = vec![C::identity(); g.len()];
    let rc = unsafe { ffi::h2hip_g_to_lagrange_bn254(g.as_ptr() as *const u64, k, out.as_mut_ptr() as *mut u64) };
    if rc != 0 {
        return None;
    }
    Some(out)
}

/// Whether the engine takes a transform of 2^log_n elements of `G` at all (callers that would allocate for the engine ask first).
pub fn takes_fft<G: 'static>(log_n: u32) -> bool {
    is::<G, Fr>() && log_n != 0 && log_n <= Fr::S && log_n >= unsafe { ffi::h2hip_ntt_min_log_n() } && layout_ok()
}

/// The guards `try_fft` applies, for every transform wrapper: G = S = bn256::Fr, 1 <= log_n <= Fr::S and at least the engine's
/// threshold, and `omega` of exact order 2^log_n (the engine's parity is defined for nothing else; the C side rejects a
/// non-reduced element but cannot know the order the caller meant).
fn fft_guards<G: 'static, S: 'static>(omega: &S, log_n: u32) -> Option<&Fr> {
    if !is::<S, Fr>() || !takes_fft::<G>(log_n) {
        return None;
    }
    let w: &Fr = unsafe { &*(omega as *const S as *const Fr) };
    if w.pow_vartime(&[1u64 << (log_n - 1)]) != -Fr::one() {
        return None;
    }
    Some(w)
}

fn fr_ptr<S>(x: &S) -> *const u64 {
    x as *const S as *const u64
}

/// `EvaluationDomain::ifft` (poly/domain.rs:353-361) in one device round trip: NTT with `omega_inv`, scaled by `divisor`.
pub fn try_ifft<G: 'static, S: 'static>(a: &mut [G], omega_inv: &S, log_n: u32, divisor: &S) -> bool {
    if log_n >= usize::BITS || a.len() != 1usize << log_n || fft_guards::<G, S>(omega_inv, log_n).is_none() {
        return false;
    }
    unsafe { ffi::h2hip_ifft_bn254_fr(a.as_mut_ptr() as *mut u64, fr_ptr(omega_inv), log_n, fr_ptr(divisor)) == 0 }
}

/// `ifft` for several columns of one size in one call: upload i + 1, transform i and download i - 1 overlap (h2hip_ifft_bn254_fr_batch).
pub fn try_ifft_batch<G: 'static, S: 'static>(columns: &mut [&mut [G]], omega_inv: &S, log_n: u32, divisor: &S) -> bool {
    if columns.is_empty() || log_n >= usize::BITS || columns.iter().any(|c| c.len() != 1usize << log_n) || fft_guards::<G, S>(omega_inv, log_n).is_none() {
        return false;
    }
    let ptrs: Vec<*mut u64> = columns.iter_mut().map(|c| c.as_mut_ptr() as *mut u64).collect();
    unsafe { ffi::h2hip_ifft_bn254_fr_batch(ptrs.as_ptr(), ptrs.len(), fr_ptr(omega_inv), log_n, fr_ptr(divisor)) == 0 }
}

/// `EvaluationDomain::coeff_to_extended` (poly/domain.rs:240-254): zero-pad, distribute powers of zeta, extended NTT.
/// `a` holds 2^k coefficients, `out` receives 2^extended_k evaluations.
pub fn try_coeff_to_extended<G: 'static, S: 'static>(a: &[G], k: u32, out: &mut [G], extended_k: u32, extended_omega: &S, g_coset: &S, g_coset_inv: &S) -> bool {
    if k > extended_k || extended_k >= usize::BITS || a.len() != 1usize << k || out.len() != 1usize << extended_k {
        return false;
    }
    if fft_guards::<G, S>(extended_omega, extended_k).is_none() {
        return false;
    }
    unsafe {
        ffi::h2hip_coeff_to_extended_bn254_fr(a.as_ptr() as *const u64, k, out.as_mut_ptr() as *mut u64, extended_k, fr_ptr(extended_omega), fr_ptr(g_coset), fr_ptr(g_coset_inv)) == 0
    }
}

/// The same in place, as the reference does it: `a` has been resized to 2^extended_k elements, of which the first 2^k are the
/// coefficients (the engine reads only those and writes all 2^extended_k).
pub fn try_coeff_to_extended_in_place<G: 'static, S: 'static>(a: &mut [G], k: u32, extended_k: u32, extended_omega: &S, g_coset: &S, g_coset_inv: &S) -> bool {
    if k > extended_k || extended_k >= usize::BITS || a.len() != 1usize << extended_k || fft_guards::<G, S>(extended_omega, extended_k).is_none() {
        return false;
    }
    let p = a.as_mut_ptr() as *mut u64;
    unsafe { ffi::h2hip_coeff_to_extended_bn254_fr(p as *const u64, k, p, extended_k, fr_ptr(extended_omega), fr_ptr(g_coset), fr_ptr(g_coset_inv)) == 0 }
}

/// `coeff_to_extended` for several polynomials in one pipelined call (plonk/evaluation.rs:306-323 extends every advice and
/// instance column): `a[i]` holds 2^k coefficients, `out[i]` receives 2^extended_k evaluations.
pub fn try_coeff_to_extended_batch<G: 'static, S: 'static>(a: &[&[G]], k: u32, out: &mut [&mut [G]], extended_k: u32, extended_omega: &S, g_coset: &S, g_coset_inv: &S) -> bool {
    if a.is_empty() || a.len() != out.len() || k > extended_k || extended_k >= usize::BITS {
        return false;
    }
    if a.iter().any(|c| c.len() != 1usize << k) || out.iter().any(|c| c.len() != 1usize << extended_k) || fft_guards::<G, S>(extended_omega, extended_k).is_none() {
        return false;
    }
    let ins: Vec<*const u64> = a.iter().map(|c| c.as_ptr() as *const u64).collect();
    let outs: Vec<*mut u64> = out.iter_mut().map(|c| c.as_mut_ptr() as *mut u64).collect();
    unsafe {
        ffi::h2hip_coeff_to_extended_bn254_fr_batch(ins.as_ptr(), k, outs.as_ptr(), ins.len(), extended_k, fr_ptr(extended_omega), fr_ptr(g_coset), fr_ptr(g_coset_inv)) == 0
    }
}

/// `EvaluationDomain::extended_to_coeff` (poly/domain.rs:281-303) before its `truncate`.
pub fn try_extended_to_coeff<G: 'static, S: 'static>(a: &mut [G], extended_k: u32, extended_omega_inv: &S, extended_ifft_divisor: &S, g_coset: &S, g_coset_inv: &S) -> bool {
    if extended_k >= usize::BITS || a.len() != 1usize << extended_k || fft_guards::<G, S>(extended_omega_inv, extended_k).is_none() {
        return false;
    }
    unsafe {
        ffi::h2hip_extended_to_coeff_bn254_fr(a.as_mut_ptr() as *mut u64, extended_k, fr_ptr(extended_omega_inv), fr_ptr(extended_ifft_divisor), fr_ptr(g_coset), fr_ptr(g_coset_inv)) == 0
    }
}
