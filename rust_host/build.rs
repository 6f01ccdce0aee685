// Link against libsrhip.so (C ABI in include/srhip.h).  SRHIP_LIB_DIR names the directory that
// holds it; the default is the in-tree build location.
use std::env;
use std::path::PathBuf;

fn main() {
    let dir = env::var("SRHIP_LIB_DIR").map(PathBuf::from).unwrap_or_else(|_| {
        PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("..").join("rusty_sr_amd")
    });
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=srhip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=SRHIP_LIB_DIR");
}
