// Links librsx.so (built by `python -m radix_sort_amd._build`, i.e. one hipcc invocation).
fn main() {
    let dir = std::env::var("RSX_LIB_DIR").unwrap_or_else(|_| "../radix_sort_amd/lib".to_string());
    println!("cargo:rustc-link-search=native={dir}");
    println!("cargo:rustc-link-lib=dylib=rsx");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{dir}");
    println!("cargo:rerun-if-env-changed=RSX_LIB_DIR");
}
