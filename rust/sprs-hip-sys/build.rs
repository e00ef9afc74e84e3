// Same convention as suitesparse_bindings/suitesparse_ldl_sys/build.rs:3-9.
fn main() {
    if let Ok(dir) = std::env::var("SPRS_HIP_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
    }
    println!("cargo:rustc-link-lib=dylib=sprs_hip");
    println!("cargo:rerun-if-env-changed=SPRS_HIP_LIB_DIR");
}
