// Links librustracer_hip.so (built by `make -C rustracer_amd/csrc`).  RUSTRACER_HIP_LIB_DIR names the directory that holds it;
// the default is the in-tree location relative to this crate.
use std::env;
use std::path::PathBuf;

fn main() {
    let dir = env::var("RUSTRACER_HIP_LIB_DIR")
        .map(PathBuf::from)
        .unwrap_or_else(|_| PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("../rustracer_amd"));
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=rustracer_hip");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rerun-if-env-changed=RUSTRACER_HIP_LIB_DIR");
    println!("cargo:rerun-if-changed=../include/rustracer_hip.h");
}
