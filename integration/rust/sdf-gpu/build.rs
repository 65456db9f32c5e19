// Link against libsdfgpu.so.  SDFGPU_LIB_DIR = directory that holds it
// (…/sdf_amd after `make -C sdf_amd/csrc`).
fn main() {
    let dir = std::env::var("SDFGPU_LIB_DIR").expect("set SDFGPU_LIB_DIR to the directory of libsdfgpu.so");
    println!("cargo:rustc-link-search=native={}", dir);
    println!("cargo:rustc-link-lib=dylib=sdfgpu");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir);
    println!("cargo:rerun-if-env-changed=SDFGPU_LIB_DIR");
}
