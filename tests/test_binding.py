"""CPU: the Rust binding shipped as files (halo2hip-sys/, patches/) stays consistent with the C ABI and with the
reference tree it patches.  No Rust toolchain exists in this image, so nothing here compiles Rust:
  * halo2hip-sys/src/ffi.rs is the translation of include/halo2hip.h by tools/gen_rust_extern.py -- regenerated and compared;
  * every `extern "C"` name is exported by libhalo2hip.so, with the header's arity;
  * the #[repr(C)] structs of halo2hip-sys/src/evalh.rs list the header's fields, in the header's order, with matching types;
  * patches/*.patch apply cleanly (dry run) to a scratch copy of the reference files they touch (skipped where
    /root/reference does not exist, i.e. on the GPU box)."""
import os
import re
import shutil
import subprocess
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_rust_extern  # noqa: E402

SYS = os.path.join(ROOT, "halo2hip-sys")
REF = "/root/reference"


def test_ffi_rs_is_the_translation_of_the_header():
    assert open(os.path.join(SYS, "src", "ffi.rs")).read() == gen_rust_extern.render()


def _rust_externs():
    text = open(os.path.join(SYS, "src", "ffi.rs")).read()
    out = {}
    for name, args in re.findall(r"pub fn (h2hip_[a-z0-9_]+)\(([^)]*)\)", text):
        out[name] = [a for a in (x.strip() for x in args.split(",")) if a]
    return out


def test_every_extern_is_exported_with_the_headers_arity(h2):
    externs = _rust_externs()
    header = {name: args for _, name, args in gen_rust_extern.prototypes()}
    assert set(externs) == set(header) and len(externs) > 40
    L = h2.lib()
    for name, args in externs.items():
        assert hasattr(L, name), "libhalo2hip.so does not export " + name
        assert len(args) == len(header[name]), name
    # the drop-in's own calls are among them
    lib_rs = open(os.path.join(SYS, "src", "lib.rs")).read() + open(os.path.join(SYS, "src", "evalh.rs")).read()
    used = set(re.findall(r"ffi::(h2hip_[a-z0-9_]+)", lib_rs))
    assert {"h2hip_msm_bn254", "h2hip_ntt_bn254_fr", "h2hip_bases_pin", "h2hip_bases_unpin", "h2hip_evaluate_h_bn254"} <= used <= set(externs)


C2R = {"uint32_t": "u32", "int32_t": "i32", "const uint64_t*": "*const u64", "const int32_t*": "*const i32", "const uint32_t*": "*const u32",
       "const uint64_t* const*": "*const *const u64", "const h2hip_calculation*": "*const h2hip_calculation",
       "const h2hip_value_source*": "*const h2hip_value_source", "const h2hip_graph*": "*const h2hip_graph",
       "h2hip_value_source": "h2hip_value_source", "h2hip_graph": "h2hip_graph"}


def _c_structs():
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "halo2hip.h")).read(), flags=re.S)
    out = {}
    for body, name in re.findall(r"typedef struct \{(.*?)\}\s*(h2hip_[a-z_]+);", text, flags=re.S):
        fields = []
        for decl in body.split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            parts = [x.strip() for x in decl.split(",")]
            m = re.match(r"^(.*?)([a-z_0-9]+)$", parts[0])
            first_type, first_name = m.group(1).strip(), m.group(2)
            if len(parts) == 1:
                fields.append((first_name, C2R[first_type.replace(" *", "*")]))
                continue
            base = first_type.rstrip("* ")  # `const uint64_t *a, *b`: the stars belong to the declarators
            for k, part in enumerate(parts):
                decl_k = (first_type[len(base):] + first_name) if k == 0 else part
                decl_k = decl_k.replace(" ", "")
                stars = len(decl_k) - len(decl_k.lstrip("*"))
                fields.append((decl_k.lstrip("*"), C2R[base + "*" * stars]))
        out[name] = fields
    return out


def _rust_structs():
    text = re.sub(r"//[^\n]*", "", open(os.path.join(SYS, "src", "evalh.rs")).read())
    out = {}
    for name, body in re.findall(r"#\[repr\(C\)\][^{]*?pub struct (h2hip_[a-z_]+) \{(.*?)\n\}", text, flags=re.S):
        out[name] = [(n, " ".join(t.split())) for n, t in re.findall(r"pub ([a-z_0-9]+): ([^,\n]+),", body)]
    return out


def test_repr_c_structs_match_the_header_field_for_field():
    c, r = _c_structs(), _rust_structs()
    assert set(c) == {"h2hip_value_source", "h2hip_calculation", "h2hip_graph", "h2hip_evalh_desc"} == set(r)
    for name in c:
        assert r[name] == c[name], name


def test_enum_constants_match_the_header():
    header = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "halo2hip.h")).read(), flags=re.S)
    rust = open(os.path.join(SYS, "src", "evalh.rs")).read()
    c_vals = dict(re.findall(r"(H2HIP_(?:VS|CALC|ANY)_[A-Z_]+) = (\d+)", header))
    r_vals = dict(re.findall(r"pub const (H2HIP_(?:VS|CALC|ANY)_[A-Z_]+): u32 = (\d+);", rust))
    assert c_vals == r_vals and len(c_vals) == 11 + 8 + 3
    lib = open(os.path.join(SYS, "src", "lib.rs")).read()
    for code, val in re.findall(r"#define (H2HIP_(?:OK|EINVAL|EDEVICE|ENOMEM)) (\d+)", header):
        assert re.search(r"pub const %s: i32 = %s;" % (code, val), lib), code


def test_crate_sources_are_complete():
    """no elisions, no placeholders: VERDICT r2 asked for files a maintainer can build, not snippets"""
    for rel in ("Cargo.toml", "build.rs", "src/lib.rs", "src/ffi.rs", "src/evalh.rs"):
        txt = open(os.path.join(SYS, rel)).read()
        assert "..." not in txt.replace("...)", "") and "todo!" not in txt and "unimplemented!" not in txt, rel
    cargo = open(os.path.join(SYS, "Cargo.toml")).read()
    assert 'links = "halo2hip"' in cargo and 'tag = "0.3.1"' in cargo  # the curve crate at the reference's own pin
    assert "rustc-link-lib=dylib=halo2hip" in open(os.path.join(SYS, "build.rs")).read()


@pytest.mark.skipif(not os.path.isdir(REF) or shutil.which("patch") is None, reason="needs the reference tree and patch(1)")
def test_patches_apply_to_the_reference(tmp_path):
    patches = sorted(f for f in os.listdir(os.path.join(ROOT, "patches")) if f.endswith(".patch"))
    assert len(patches) >= 6
    touched = set()
    for p in patches:
        for line in open(os.path.join(ROOT, "patches", p)):
            if line.startswith("+++ b/"):
                touched.add(line[6:].strip())
    assert {"halo2_proofs/src/arithmetic.rs", "halo2_proofs/src/poly/kzg/commitment.rs", "halo2_proofs/src/plonk/evaluation.rs",
            "halo2_proofs/Cargo.toml", "halo2_proofs/src/poly/domain.rs", "halo2_proofs/src/plonk/prover.rs",
            "halo2_proofs/src/poly/commitment.rs", "halo2_proofs/src/plonk/vanishing/prover.rs", "halo2_proofs/src/plonk.rs"} <= touched
    for rel in touched:  # a scratch copy of just those files (nothing of the reference enters the repository)
        dst = tmp_path / rel
        dst.parent.mkdir(parents=True, exist_ok=True)
        shutil.copy(os.path.join(REF, rel), dst)
    for p in patches:
        r = subprocess.run(["patch", "-p1", "--dry-run", "--fuzz=0", "-i", os.path.join(ROOT, "patches", p)], cwd=tmp_path, capture_output=True, text=True)
        assert r.returncode == 0, p + "\n" + r.stdout + r.stderr
    # and for real, so that the dispatch lines can be seen where DESIGN.md says they land
    for p in patches:
        subprocess.run(["patch", "-p1", "--fuzz=0", "-i", os.path.join(ROOT, "patches", p)], cwd=tmp_path, check=True, capture_output=True)
    arith = (tmp_path / "halo2_proofs/src/arithmetic.rs").read_text().splitlines()
    i = next(k for k, ln in enumerate(arith) if ln.startswith("pub fn best_multiexp"))
    assert "assert_eq!(coeffs.len(), bases.len());" in arith[i + 1] and "halo2hip_sys::try_multiexp::<C>(coeffs, bases)" in arith[i + 2]
    j = next(k for k, ln in enumerate(arith) if "assert_eq!(n, 1 << log_n);" in ln)
    assert "halo2hip_sys::try_fft(a, &omega, log_n)" in arith[j + 1]
    # 0004: the domain conversions try the fused engine calls first, and the two places that convert columns back to back batch them
    dom = (tmp_path / "halo2_proofs/src/poly/domain.rs").read_text()
    for call in ("halo2hip_sys::try_ifft(a, &omega_inv, log_n, &divisor)", "halo2hip_sys::try_coeff_to_extended_in_place(", "halo2hip_sys::try_extended_to_coeff(",
                 "halo2hip_sys::try_ifft_batch(", "halo2hip_sys::try_coeff_to_extended_batch(", "pub fn lagrange_to_coeff_batch(", "pub fn coeff_to_extended_batch("):
        assert call in dom, call
    prover = (tmp_path / "halo2_proofs/src/plonk/prover.rs").read_text()
    assert "domain.lagrange_to_coeff_batch(advice_polys)" in prover and ".map(|poly| domain.lagrange_to_coeff(poly))" not in prover
    evaluation = (tmp_path / "halo2_proofs/src/plonk/evaluation.rs").read_text()
    assert evaluation.count("domain.coeff_to_extended_batch(") == 2 and "domain.coeff_to_extended(poly.clone())" not in evaluation
    # 0005: the back-to-back commits go through the batch methods, which ParamsKZG overrides with try_multiexp_batch
    assert "params.commit_lagrange_batch(&advice_values, &blinds)" in prover and "params.commit_lagrange(poly, *blind)" not in prover
    assert "params.commit_batch(&h_pieces, &h_blinds)" in (tmp_path / "halo2_proofs/src/plonk/vanishing/prover.rs").read_text()
    kzg = (tmp_path / "halo2_proofs/src/poly/kzg/commitment.rs").read_text()
    assert kzg.count("halo2hip_sys::try_multiexp_batch::<E::G1Affine>(") == 2
    trait = (tmp_path / "halo2_proofs/src/poly/commitment.rs").read_text()
    assert "fn commit_lagrange_batch(" in trait and "fn commit_batch(" in trait
    # 0006: the proving key releases the device copies of its constant columns
    plonk = (tmp_path / "halo2_proofs/src/plonk.rs").read_text()
    assert "impl<C: CurveAffine> Drop for ProvingKey<C>" in plonk and "halo2hip_sys::evalh::unpin_key_columns(&columns)" in plonk
    # every halo2hip_sys item the patches call exists in the crate
    called = set()
    for p in patches:
        called |= set(re.findall(r"halo2hip_sys::((?:evalh::)?[a-z_A-Z0-9]+)", open(os.path.join(ROOT, "patches", p)).read()))
    lib = open(os.path.join(SYS, "src", "lib.rs")).read()
    ev = open(os.path.join(SYS, "src", "evalh.rs")).read()
    for item in called - {"evalh"}:
        src, name = (ev, item[7:]) if item.startswith("evalh::") else (lib, item)
        assert re.search(r"pub (?:unsafe )?(?:fn|struct|const|mod) %s\b" % re.escape(name), src), item


def test_integration_appendix_is_current(tmp_path):
    """INTEGRATION.md's appendix lists every entry point include/halo2hip.h declares, at the line it is declared on: regenerating it
    (tools/gen_integration_appendix.py) changes nothing."""
    import re
    import shutil
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    text = open(os.path.join(root, "INTEGRATION.md")).read()
    header = open(os.path.join(root, "include", "halo2hip.h")).read()
    for name in sorted(set(re.findall(r"^(?:int|void|size_t|uint32_t|uint64_t|const char\s*\*)\s+\*?\s*(h2hip_[a-z0-9_]+)\s*\(", header, flags=re.M))):
        assert "| `%s` |" % name in text, name
    scratch = tmp_path / "repo"
    for sub in ("include", "tools"):
        shutil.copytree(os.path.join(root, sub), scratch / sub, ignore=shutil.ignore_patterns("__pycache__", "*.o", "instr_rate", "selftest", "pcie_rate", "graph_chain", "atomic_rate"))
    shutil.copy(os.path.join(root, "INTEGRATION.md"), scratch / "INTEGRATION.md")
    subprocess.check_call([sys.executable, str(scratch / "tools" / "gen_integration_appendix.py")])
    assert open(scratch / "INTEGRATION.md").read() == text, "INTEGRATION.md's appendix is stale: run tools/gen_integration_appendix.py"
