// Link against libhsw.so built by `make -C halo2-dynamic-sha256_amd/csrc`.
fn main() {
    let dir = std::env::var("HSW_LIB_DIR").unwrap_or_else(|_| "../../halo2-dynamic-sha256_amd".to_string());
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=hsw");
    println!("cargo:rerun-if-env-changed=HSW_LIB_DIR");
}
