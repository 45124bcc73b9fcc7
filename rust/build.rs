// UNVERIFIED (no Rust toolchain in the build image).  build.rs for the rsparse crate root: builds librsparse_hip.so with
// hipcc for gfx950 when RSPARSE_HIP_SRC points at this repository, and links it.
use std::process::Command;

fn main() {
    let dir = match std::env::var("RSPARSE_HIP_DIR") {
        Ok(d) => d, // a directory that already holds librsparse_hip.so
        Err(_) => {
            let src = std::env::var("RSPARSE_HIP_SRC").expect("set RSPARSE_HIP_DIR (prebuilt) or RSPARSE_HIP_SRC (this repository)");
            let ok = Command::new("make").arg("-C").arg(format!("{src}/rsparse_amd/csrc")).status().expect("make").success();
            assert!(ok, "hipcc build of librsparse_hip.so failed");
            format!("{src}/rsparse_amd")
        }
    };
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=rsparse_hip");
    println!("cargo:rerun-if-env-changed=RSPARSE_HIP_DIR");
    println!("cargo:rerun-if-env-changed=RSPARSE_HIP_SRC");
}
