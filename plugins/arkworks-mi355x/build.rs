// SOURCE ONLY (see Cargo.toml).  Tells rustc where libzl_backend.so lives.
fn main() {
    if let Ok(dir) = std::env::var("ZL_BACKEND_LIB_DIR") {
        println!("cargo:rustc-link-search=native={dir}");
        println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    }
    println!("cargo:rerun-if-env-changed=ZL_BACKEND_LIB_DIR");
}
