// Links libaprilgrid_amd.so (make -C aprilgrid-rs_amd).  AGX_LIB_DIR names the directory that holds it; by default the
// repository's own build output, two levels up.
fn main() {
    let dir = std::env::var("AGX_LIB_DIR").unwrap_or_else(|_| {
        let here = std::env::var("CARGO_MANIFEST_DIR").unwrap();
        format!("{here}/../../aprilgrid-rs_amd")
    });
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=aprilgrid_amd");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=AGX_LIB_DIR");
}
