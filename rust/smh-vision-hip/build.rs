fn main() {
    // libsmh_vision_hip.so built from this repository (make -C squad-mortar-helper_amd/csrc)
    println!("cargo:rustc-link-search=native={}", std::env::var("SMH_VISION_HIP_DIR").unwrap());
    println!("cargo:rustc-link-lib=dylib=smh_vision_hip");
}
