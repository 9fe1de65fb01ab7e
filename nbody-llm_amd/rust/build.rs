// Links the prebuilt C-ABI library (make -C nbody-llm_amd/csrc).
fn main() {
    let dir = std::env::var("NBODY_HIP_LIB_DIR").unwrap_or_else(|_| "..".to_string());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=nbody_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
}
