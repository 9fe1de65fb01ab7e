// Links libnbody_hip.so (built by `make -C nbody-llm_amd/csrc`).  NBODY_HIP_LIB_DIR overrides the search path.
fn main() {
    let dir = std::env::var("NBODY_HIP_LIB_DIR").unwrap_or_else(|_| {
        let manifest = std::env::var("CARGO_MANIFEST_DIR").unwrap();
        format!("{manifest}/..")
    });
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=nbody_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=NBODY_HIP_LIB_DIR");
}
