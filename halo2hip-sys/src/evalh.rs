//! `Evaluator::evaluate_h` (halo2_proofs/src/plonk/evaluation.rs:280-522) through the engine: the `#[repr(C)]` mirrors of
//! include/halo2hip.h's structures, and an owning builder (`FlatGraph`) that halo2_proofs fills from its private
//! `GraphEvaluator` (patches/0003-evaluate-h.patch adds the two `flat()` methods that do so).
use std::os::raw::c_int;

// ---- include/halo2hip.h: enums ------------------------------------------------------------------------------------
pub const H2HIP_VS_CONSTANT: u32 = 0;
pub const H2HIP_VS_INTERMEDIATE: u32 = 1;
pub const H2HIP_VS_FIXED: u32 = 2;
pub const H2HIP_VS_ADVICE: u32 = 3;
pub const H2HIP_VS_INSTANCE: u32 = 4;
pub const H2HIP_VS_CHALLENGE: u32 = 5;
pub const H2HIP_VS_BETA: u32 = 6;
pub const H2HIP_VS_GAMMA: u32 = 7;
pub const H2HIP_VS_THETA: u32 = 8;
pub const H2HIP_VS_Y: u32 = 9;
pub const H2HIP_VS_PREVIOUS: u32 = 10;

pub const H2HIP_CALC_ADD: u32 = 0;
pub const H2HIP_CALC_SUB: u32 = 1;
pub const H2HIP_CALC_MUL: u32 = 2;
pub const H2HIP_CALC_SQUARE: u32 = 3;
pub const H2HIP_CALC_DOUBLE: u32 = 4;
pub const H2HIP_CALC_NEGATE: u32 = 5;
pub const H2HIP_CALC_HORNER: u32 = 6;
pub const H2HIP_CALC_STORE: u32 = 7;

pub const H2HIP_ANY_ADVICE: u32 = 0;
pub const H2HIP_ANY_FIXED: u32 = 1;
pub const H2HIP_ANY_INSTANCE: u32 = 2;

// ---- include/halo2hip.h: structures (field for field, same order) ----------------------------------------------------
/// `ValueSource` (evaluation.rs:37-60): `kind` = H2HIP_VS_*, `a` = constant / intermediate / challenge / column index,
/// `b` = index into the graph's rotations for Fixed / Advice / Instance.
#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq, Eq)]
pub struct h2hip_value_source {
    pub kind: u32,
    pub a: u32,
    pub b: u32,
}

/// `Calculation` + `CalculationInfo` (evaluation.rs:108-127, :213-219).  Horner: `x` = start value, `y` = factor, the
/// parts are `h2hip_graph::parts[parts_offset .. parts_offset + parts_count]`.
#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct h2hip_calculation {
    pub op: u32,
    pub target: u32,
    pub x: h2hip_value_source,
    pub y: h2hip_value_source,
    pub parts_offset: u32,
    pub parts_count: u32,
}

/// `GraphEvaluator` (evaluation.rs:191-201); borrowed view of a `FlatGraph`.
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct h2hip_graph {
    pub constants: *const u64, // n_constants x 4 limbs, Montgomery form
    pub n_constants: u32,
    pub rotations: *const i32,
    pub n_rotations: u32,
    pub calculations: *const h2hip_calculation,
    pub n_calculations: u32,
    pub parts: *const h2hip_value_source,
    pub n_parts: u32,
    pub num_intermediates: u32,
}

/// Everything one circuit instance of `evaluate_h` reads.  Pointers are host pointers for `h2hip_evaluate_h_bn254`.
#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct h2hip_evalh_desc {
    pub k: u32,
    pub extended_k: u32,
    pub extended_omega: *const u64,
    pub g_coset: *const u64,
    pub g_coset_inv: *const u64,
    pub n_fixed: u32,
    pub n_advice: u32,
    pub n_instance: u32,
    pub n_challenges: u32,
    pub fixed_cosets: *const *const u64,
    pub advice_polys: *const *const u64,
    pub instance_polys: *const *const u64,
    pub challenges: *const u64,
    pub y: *const u64,
    pub beta: *const u64,
    pub gamma: *const u64,
    pub theta: *const u64,
    pub l0: *const u64,
    pub l_last: *const u64,
    pub l_active_row: *const u64,
    pub custom_gates: h2hip_graph,
    pub n_perm_sets: u32,
    pub n_perm_columns: u32,
    pub chunk_len: u32,
    pub last_rotation: i32,
    pub perm_product_cosets: *const *const u64,
    pub perm_column_kind: *const u32,
    pub perm_column_index: *const u32,
    pub perm_cosets: *const *const u64,
    pub zeta: *const u64,
    pub delta: *const u64,
    pub n_lookups: u32,
    pub lookup_graphs: *const h2hip_graph,
    pub lookup_product_polys: *const *const u64,
    pub lookup_permuted_input_polys: *const *const u64,
    pub lookup_permuted_table_polys: *const *const u64,
}

// ---- owning builder -------------------------------------------------------------------------------------------------
/// A `GraphEvaluator` flattened into plain vectors.  halo2_proofs builds one per graph once per proving key; `view()`
/// lends it to the engine for the duration of a call.
#[derive(Clone, Debug, Default)]
pub struct FlatGraph {
    pub constants: Vec<[u64; 4]>,
    pub rotations: Vec<i32>,
    pub calculations: Vec<h2hip_calculation>,
    pub parts: Vec<h2hip_value_source>,
    pub num_intermediates: u32,
}

impl FlatGraph {
    pub fn vs(kind: u32, a: usize, b: usize) -> h2hip_value_source {
        h2hip_value_source { kind, a: a as u32, b: b as u32 }
    }

    /// a calculation with at most two operands (everything but Horner)
    pub fn push(&mut self, op: u32, target: usize, x: h2hip_value_source, y: h2hip_value_source) {
        self.calculations.push(h2hip_calculation { op, target: target as u32, x, y, parts_offset: 0, parts_count: 0 });
    }

    /// `Calculation::Horner(start, parts, factor)` (evaluation.rs:122-123)
    pub fn push_horner(&mut self, target: usize, start: h2hip_value_source, parts: &[h2hip_value_source], factor: h2hip_value_source) {
        let parts_offset = self.parts.len() as u32;
        self.parts.extend_from_slice(parts);
        self.calculations.push(h2hip_calculation {
            op: H2HIP_CALC_HORNER,
            target: target as u32,
            x: start,
            y: factor,
            parts_offset,
            parts_count: parts.len() as u32,
        });
    }

    /// borrowed C view; valid while `self` is neither moved nor modified
    pub fn view(&self) -> h2hip_graph {
        h2hip_graph {
            constants: self.constants.as_ptr() as *const u64,
            n_constants: self.constants.len() as u32,
            rotations: self.rotations.as_ptr(),
            n_rotations: self.rotations.len() as u32,
            calculations: self.calculations.as_ptr(),
            n_calculations: self.calculations.len() as u32,
            parts: self.parts.as_ptr(),
            n_parts: self.parts.len() as u32,
            num_intermediates: self.num_intermediates,
        }
    }
}

impl FlatGraph {
    /// append a constant (`GraphEvaluator::constants`); false when `F` is not bn256::Fr
    pub fn push_constant<F: 'static>(&mut self, c: &F) -> bool {
        match limbs_of(c) {
            Some(l) => {
                self.constants.push(l);
                true
            }
            None => false,
        }
    }
}

/// the 4 Montgomery limbs of a bn256::Fr (None for any other type)
pub fn limbs_of<F: 'static>(x: &F) -> Option<[u64; 4]> {
    if std::any::TypeId::of::<F>() != std::any::TypeId::of::<halo2curves::bn256::Fr>() || std::mem::size_of::<F>() != 32 {
        return None;
    }
    Some(unsafe { std::mem::transmute_copy::<F, [u64; 4]>(x) })
}

/// One circuit instance of `evaluate_h` described with borrowed slices: what halo2_proofs (which forbids unsafe code) hands
/// over.  Column slices are the `Polynomial`s' value vectors (`Deref<Target = [F]>`, poly.rs:116-128).
pub struct EvalHInput<'a, F> {
    pub k: u32,
    pub extended_k: u32,
    pub extended_omega: F,
    pub g_coset: F,
    pub g_coset_inv: F,
    /// pk.fixed_cosets (2^extended_k each)
    pub fixed_cosets: Vec<&'a [F]>,
    /// advice / instance polynomials in coefficient form (2^k each); the engine forms their cosets (evaluation.rs:306-323)
    pub advice_polys: Vec<&'a [F]>,
    pub instance_polys: Vec<&'a [F]>,
    pub challenges: &'a [F],
    pub y: F,
    pub beta: F,
    pub gamma: F,
    pub theta: F,
    /// pk.l0 / l_last / l_active_row (2^extended_k each)
    pub l0: &'a [F],
    pub l_last: &'a [F],
    pub l_active_row: &'a [F],
    pub custom_gates: &'a FlatGraph,
    /// sets[i].permutation_product_coset (2^extended_k each); empty skips the permutation argument
    pub perm_product_cosets: Vec<&'a [F]>,
    /// (H2HIP_ANY_*, column index) of cs.permutation.columns[j]
    pub perm_columns: Vec<(u32, u32)>,
    /// pk.permutation.cosets[j] (2^extended_k each)
    pub perm_cosets: Vec<&'a [F]>,
    /// cs.degree() - 2
    pub chunk_len: u32,
    /// -(cs.blinding_factors() + 1)
    pub last_rotation: i32,
    pub zeta: F,
    pub delta: F,
    pub lookup_graphs: &'a [FlatGraph],
    /// lookup.product_poly / permuted_input_poly / permuted_table_poly, coefficient form (2^k each)
    pub lookup_product_polys: Vec<&'a [F]>,
    pub lookup_permuted_input_polys: Vec<&'a [F]>,
    pub lookup_permuted_table_polys: Vec<&'a [F]>,
}

/// Safe front of `h2hip_evaluate_h_bn254`: checks the element type and every slice length, builds the pointer tables, calls
/// the engine.  `values` (2^extended_k elements) is read and written.  false: the engine declined, run the CPU body.
pub fn try_evaluate_h<F: 'static + Copy>(inp: &EvalHInput<'_, F>, values: &mut [F]) -> bool {
    if std::any::TypeId::of::<F>() != std::any::TypeId::of::<halo2curves::bn256::Fr>() || std::mem::size_of::<F>() != 32 {
        return false;
    }
    let n = 1usize << inp.k;
    let en = 1usize << inp.extended_k;
    let all_len = |v: &Vec<&[F]>, len: usize| v.iter().all(|s| s.len() == len);
    let n_lookups = inp.lookup_graphs.len();
    if values.len() != en
        || !all_len(&inp.fixed_cosets, en)
        || !all_len(&inp.advice_polys, n)
        || !all_len(&inp.instance_polys, n)
        || inp.l0.len() != en
        || inp.l_last.len() != en
        || inp.l_active_row.len() != en
        || !all_len(&inp.perm_product_cosets, en)
        || !all_len(&inp.perm_cosets, en)
        || inp.perm_cosets.len() != inp.perm_columns.len()
        || !all_len(&inp.lookup_product_polys, n)
        || !all_len(&inp.lookup_permuted_input_polys, n)
        || !all_len(&inp.lookup_permuted_table_polys, n)
        || inp.lookup_product_polys.len() != n_lookups
        || inp.lookup_permuted_input_polys.len() != n_lookups
        || inp.lookup_permuted_table_polys.len() != n_lookups
    {
        return false;
    }
    let table = |v: &Vec<&[F]>| -> Vec<*const u64> { v.iter().map(|s| s.as_ptr() as *const u64).collect() };
    let fixed = table(&inp.fixed_cosets);
    let advice = table(&inp.advice_polys);
    let instance = table(&inp.instance_polys);
    let perm_prod = table(&inp.perm_product_cosets);
    let perm_cosets = table(&inp.perm_cosets);
    let lk_prod = table(&inp.lookup_product_polys);
    let lk_in = table(&inp.lookup_permuted_input_polys);
    let lk_tab = table(&inp.lookup_permuted_table_polys);
    let kinds: Vec<u32> = inp.perm_columns.iter().map(|c| c.0).collect();
    let indices: Vec<u32> = inp.perm_columns.iter().map(|c| c.1).collect();
    let lookup_views: Vec<h2hip_graph> = inp.lookup_graphs.iter().map(|g| g.view()).collect();
    let p = |x: &F| x as *const F as *const u64;
    let desc = h2hip_evalh_desc {
        k: inp.k,
        extended_k: inp.extended_k,
        extended_omega: p(&inp.extended_omega),
        g_coset: p(&inp.g_coset),
        g_coset_inv: p(&inp.g_coset_inv),
        n_fixed: fixed.len() as u32,
        n_advice: advice.len() as u32,
        n_instance: instance.len() as u32,
        n_challenges: inp.challenges.len() as u32,
        fixed_cosets: fixed.as_ptr(),
        advice_polys: advice.as_ptr(),
        instance_polys: instance.as_ptr(),
        challenges: inp.challenges.as_ptr() as *const u64,
        y: p(&inp.y),
        beta: p(&inp.beta),
        gamma: p(&inp.gamma),
        theta: p(&inp.theta),
        l0: inp.l0.as_ptr() as *const u64,
        l_last: inp.l_last.as_ptr() as *const u64,
        l_active_row: inp.l_active_row.as_ptr() as *const u64,
        custom_gates: inp.custom_gates.view(),
        n_perm_sets: perm_prod.len() as u32,
        n_perm_columns: perm_cosets.len() as u32,
        chunk_len: inp.chunk_len,
        last_rotation: inp.last_rotation,
        perm_product_cosets: perm_prod.as_ptr(),
        perm_column_kind: kinds.as_ptr(),
        perm_column_index: indices.as_ptr(),
        perm_cosets: perm_cosets.as_ptr(),
        zeta: p(&inp.zeta),
        delta: p(&inp.delta),
        n_lookups: n_lookups as u32,
        lookup_graphs: lookup_views.as_ptr(),
        lookup_product_polys: lk_prod.as_ptr(),
        lookup_permuted_input_polys: lk_in.as_ptr(),
        lookup_permuted_table_polys: lk_tab.as_ptr(),
    };
    // The proving key's own columns -- pk.fixed_cosets, pk.l0 / l_last / l_active_row, pk.permutation.cosets -- do not change from proof to
    // proof: keep them in HBM across calls (idempotent; fingerprint-guarded like the pinned bases; a failure only means they are uploaded
    // per call).  `unpin_key_columns` in ProvingKey's Drop (patch 0006) releases them.
    let mut key_columns: Vec<*const u64> = fixed.clone();
    key_columns.extend_from_slice(&perm_cosets);
    key_columns.push(desc.l0);
    key_columns.push(desc.l_last);
    key_columns.push(desc.l_active_row);
    let _ = unsafe { super::ffi::h2hip_columns_pin(key_columns.as_ptr(), key_columns.len(), en) };
    let rc: c_int = unsafe { super::ffi::h2hip_evaluate_h_bn254(&desc as *const h2hip_evalh_desc, values.as_mut_ptr() as *mut u64) };
    rc == 0
}

/// Release the device copies `try_evaluate_h` keeps of a proving key's constant columns (call from `ProvingKey`'s `Drop`, before the
/// `Vec`s are freed; unknown pointers are ignored, other element types are a no-op).
pub fn unpin_key_columns<F: 'static>(columns: &[&[F]]) {
    if std::any::TypeId::of::<F>() != std::any::TypeId::of::<halo2curves::bn256::Fr>() || columns.is_empty() {
        return;
    }
    let ptrs: Vec<*const u64> = columns.iter().map(|c| c.as_ptr() as *const u64).collect();
    unsafe { super::ffi::h2hip_columns_unpin(ptrs.as_ptr(), ptrs.len()) };
}
