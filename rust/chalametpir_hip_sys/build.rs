// SOURCE ONLY -- not compile-tested.
// Links libchalamet_hip.so; CHALAMET_HIP_LIB_DIR points at the directory produced by `make -C chalametpir_amd/csrc`
// (chalametpir_amd/lib) or wherever the library was installed.
fn main() {
    if let Ok(dir) = std::env::var("CHALAMET_HIP_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    }
    println!("cargo:rustc-link-lib=dylib=chalamet_hip");
    println!("cargo:rerun-if-env-changed=CHALAMET_HIP_LIB_DIR");
}
