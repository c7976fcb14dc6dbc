// Links libhalo2hip.so.  HALO2HIP_LIB_DIR names the directory that holds it; the default is the in-tree build
// (../halo2-pse_amd, `make -C halo2-pse_amd`).  The directory is also recorded as an rpath so that a prover binary
// finds the library without LD_LIBRARY_PATH.
use std::env;
use std::path::PathBuf;

fn main() {
    let dir = match env::var("HALO2HIP_LIB_DIR") {
        Ok(d) => PathBuf::from(d),
        Err(_) => PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("..").join("halo2-pse_amd"),
    };
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=halo2hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=HALO2HIP_LIB_DIR");
    println!("cargo:rerun-if-changed=build.rs");
}
