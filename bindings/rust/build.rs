// links libsqlrs_hip.so (built by `python -m sqlrs_amd.build` into sqlrs_amd/csrc/)
fn main() {
    let dir = std::env::var("SQLRS_HIP_LIB_DIR").unwrap_or_else(|_| "../../sqlrs_amd/csrc".to_string());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=sqlrs_hip");
    println!("cargo:rerun-if-env-changed=SQLRS_HIP_LIB_DIR");
}
