// build.rs of the reference crate once the patch set is applied: link libmi_rt.so (built by
// cs397raytracingsp22_amd/csrc/build.sh).  MI_RT_LIB_DIR = the directory holding libmi_rt.so.
fn main() {
    let dir = std::env::var("MI_RT_LIB_DIR").unwrap_or_else(|_| "../cs397raytracingsp22_amd/lib".to_string());
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=mi_rt");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    println!("cargo:rerun-if-env-changed=MI_RT_LIB_DIR");
}
