fn main() {
    let dir = std::env::var("ZKMLE_AMD_LIB_DIR").expect("set ZKMLE_AMD_LIB_DIR to the directory holding libzkmle_amd.so");
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=zkmle_amd");
}
