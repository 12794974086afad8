//! Oracle pin: dumps what the REAL `CPUFallback` (vision-cpu/src/lib.rs) computes on the reference's own sample
//! screenshots, in the shape of the goldens of the MI355X build (tests/golden/manifest.json + *.golden.npz there), so that
//! `python oracle/pin/diff_pin.py pin_reference.json` can compare the two.  The MI355X build's parity tests are green against
//! a C restatement of vision-cpu; this closes the loop against vision-cpu itself.
//!
//! Install (in a checkout of WilliamVenner/squad-mortar-helper with a Rust toolchain):
//!   cp pin_goldens.rs <checkout>/vision-cpu/tests/pin_goldens.rs
//!   # vision-cpu/Cargo.toml:
//!   #   [dev-dependencies]
//!   #   image = "0.23"
//!   #   sha2 = "0.10"
//!   cargo test -p smh-vision-cpu --release --test pin_goldens -- --nocapture
//!   -> writes <checkout>/pin_reference.json
//!
//! NOT COMPILED where it was written (no Rust toolchain in that image).  It only uses the public trait surface
//! (vision-common/src/lib.rs:30-61) the way the reference's own GPU test does (vision-gpu/src/lib.rs:562-622).
use sha2::{Digest, Sha256};
use smh_vision_common::{debug::DebugView, prelude::*, Vision};
use smh_vision_cpu::CPUFallback;
use std::fmt::Write as _;

fn sha(bytes: &[u8]) -> String {
	let mut h = Sha256::new();
	h.update(bytes);
	h.finalize().iter().fold(String::new(), |mut s, b| { let _ = write!(s, "{:02x}", b); s })
}

/// f32 values as their bit patterns: the comparison is bit-exact, no decimal round trip
fn lines_json(lines: &[Line<f32>]) -> String {
	let v: Vec<String> = lines.iter().map(|l| format!("[{},{},{},{}]", l.p0.x.to_bits(), l.p0.y.to_bits(), l.p1.x.to_bits(), l.p1.y.to_bits())).collect();
	format!("[{}]", v.join(","))
}

/// channel 0 of an RGBA debug view of a grey image; RGB of an RGBA debug view of a colour image
fn gray_of(rgba: &image::RgbaImage) -> Vec<u8> { rgba.pixels().map(|p| p.0[0]).collect() }
fn rgb_of(rgba: &image::RgbaImage) -> Vec<u8> { rgba.pixels().flat_map(|p| [p.0[0], p.0[1], p.0[2]]).collect() }

#[test]
fn dump_pin() {
	let samples = std::path::Path::new(env!("CARGO_MANIFEST_DIR")).join("../vision-common/samples");
	let mut names: Vec<_> = std::fs::read_dir(&samples).unwrap().filter_map(|e| e.ok()).map(|e| e.path())
		.filter(|p| p.extension().map(|e| e == "png").unwrap_or(false)).collect();   // PNG only: lossless, decoder-independent
	names.sort();
	let mut out = String::from("{\n");
	let mut first = true;
	for path in names {
		let name = path.file_name().unwrap().to_string_lossy().to_string();
		let entry = std::panic::catch_unwind(|| {
			let image = image::open(&path).unwrap().into_bgra8();
			let (w, h) = (image.width(), image.height());
			let frame: VisionFrame = image::ImageBuffer::from_raw(w, h, image.into_raw().into_boxed_slice()).unwrap();
			let mut cpu = CPUFallback::init().unwrap();
			cpu.load_frame(frame).unwrap();
			let mut e = format!("\"W\":{},\"H\":{}", w, h);
			match cpu.crop_to_map(true).unwrap() {
				None => e.push_str(",\"map_open\":0"),
				Some((ui_gray, rect)) => {
					let _ = write!(e, ",\"map_open\":1,\"map_rect\":[{},{},{},{}]", rect[0], rect[1], rect[2], rect[3]);
					let _ = write!(e, ",\"sha_ui_gray\":\"{}\"", sha(ui_gray.as_raw()));
					let (ui_color, _) = cpu.crop_to_map(false).unwrap().unwrap();
					let _ = write!(e, ",\"sha_ui_color\":\"{}\"", sha(ui_color.as_raw()));
					let _ = write!(e, ",\"sha_brq\":\"{}\"", sha(&rgb_of(&cpu.get_debug_view(DebugView::CroppedBRQ).unwrap())));
					// scales branch (src/vision/mod.rs:131-200 order)
					let (p, n) = cpu.ocr_preprocess().unwrap();
					let _ = write!(e, ",\"sha_ocr\":\"{}\"", sha(unsafe { std::slice::from_raw_parts(p, n) }));
					let sc = cpu.find_scales_preprocess(0).unwrap();
					let _ = write!(e, ",\"sha_scales0\":\"{}\"", sha(unsafe { &*sc }.borrow().as_raw()));
					// markers branch
					cpu.isolate_map_markers().unwrap();
					let _ = write!(e, ",\"sha_isolated\":\"{}\"", sha(&rgb_of(&cpu.get_debug_view(DebugView::LSDPreprocess).unwrap())));
					cpu.mask_marker_lines().unwrap();
					let mask = gray_of(&cpu.get_debug_view(DebugView::LSDInput).unwrap());
					let idx: Vec<u8> = mask.iter().enumerate().filter(|(_, v)| **v == 255).flat_map(|(i, _)| (i as u32).to_le_bytes()).collect();
					let _ = write!(e, ",\"sha_lsd\":\"{}\",\"n_mask_px\":{},\"sha_mask_idx\":\"{}\"", sha(&mask), idx.len() / 4, sha(&idx));
					let l15 = cpu.find_marker_lines(15).unwrap();
					let l22 = cpu.find_marker_lines(22).unwrap();
					let _ = write!(e, ",\"lines\":{},\"lines_gap22\":{}", lines_json(&l15), lines_json(&l22));
				},
			}
			e
		});
		if !first { out.push_str(",\n"); }
		first = false;
		match entry {
			Ok(e) => { let _ = write!(out, " \"{}\": {{{}}}", name, e); },
			Err(_) => { let _ = write!(out, " \"{}\": {{\"panic\":1}}", name); },   // e.g. convolution.png: the bounds arithmetic underflows
		}
	}
	out.push_str("\n}\n");
	let dst = std::path::Path::new(env!("CARGO_MANIFEST_DIR")).join("../pin_reference.json");
	std::fs::write(&dst, out).unwrap();
	println!("wrote {}", dst.display());
}
