// Links against the in-tree libczk_hip.so.  CZK_LIB_DIR overrides the default location
// (<repo>/collaborative-zksnark_amd, i.e. ../../collaborative-zksnark_amd relative to this crate).
use std::env;
use std::path::PathBuf;

fn main() {
    let dir = env::var("CZK_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap())
            .join("..")
            .join("..")
            .join("collaborative-zksnark_amd")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=czk_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=CZK_LIB_DIR");
    println!("cargo:rerun-if-changed=../../include/czk.h");
}
