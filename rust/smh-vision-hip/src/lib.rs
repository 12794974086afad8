//! smh-vision-hip: the reference's `Vision` plugin trait implemented on top of libsmh_vision_hip.so (MI355X / gfx950).
//!
//! Drop-in for the `smh_vision_gpu` plugin (vision-gpu/src/lib.rs:11): copy this directory into the reference workspace as
//! `smh-vision-hip/`, add it to `[workspace] members` (Cargo.toml:8-17), build with
//! `SMH_VISION_HIP_DIR=<dir of libsmh_vision_hip.so> cargo build -p smh-vision-hip --release`, and change the plugin name in
//! `src/vision/hardware.rs:63` from "smh_vision_gpu" to "smh_vision_hip_plugin".
//!
//! NOT COMPILED in the repository this file ships in (its build image has no Rust toolchain): treat the first `cargo check`
//! as part of the integration.  The C side of every call is covered by the C / ctypes tests of that repository.
use smh_vision_common::{debug, prelude::*, Vision};
use std::{ffi::{c_char, c_int, c_void, CStr}, ptr, sync::Arc};

smh_vision_common::export_dylib_wrapper!(smh_vision_hip_plugin => HipInstance);

#[repr(C)] #[derive(Clone, Copy, Default)]
struct SmhvLine { x0: f32, y0: f32, x1: f32, y1: f32 }   // == Line<f32> #[repr(C)] (util/src/geometry.rs:169-180)

extern "C" {
    fn smhv_init(device: c_int, log: Option<extern "C" fn(c_int, *const c_char)>, out: *mut *mut c_void) -> c_int;
    fn smhv_shutdown(ctx: *mut c_void);
    fn smhv_thread_ctx(ctx: *mut c_void) -> c_int;
    fn smhv_last_error() -> *const c_char;
    fn smhv_load_frame(ctx: *mut c_void, bgra: *const u8, w: u32, h: u32) -> c_int;
    fn smhv_load_frame_view(ctx: *mut c_void, parent_bgra: *const u8, parent_w: u32, parent_h: u32, x: u32, y: u32, w: u32, h: u32) -> c_int;
    fn smhv_crop_to_map(ctx: *mut c_void, grayscale: c_int, map_open: *mut c_int, roi: *mut u32, ui_rgba: *mut u8) -> c_int;
    fn smhv_map_bounds(w: u32, h: u32, xywh: *mut u32) -> c_int;
    fn smhv_ocr_preprocess(ctx: *mut c_void, out: *mut *const u8, len: *mut usize) -> c_int;
    fn smhv_find_scales_preprocess(ctx: *mut c_void, y: u32, out: *mut *const u8, w: *mut u32, h: *mut u32) -> c_int;
    fn smhv_isolate_map_markers(ctx: *mut c_void) -> c_int;
    fn smhv_mask_marker_lines(ctx: *mut c_void) -> c_int;
    fn smhv_find_longest_line(ctx: *mut c_void, x: f32, y: f32, max_gap: f32, line: *mut SmhvLine, len_sq: *mut f32) -> c_int;
    fn smhv_find_marker_lines(ctx: *mut c_void, max_gap: u32, out: *mut SmhvLine, n: *mut u32) -> c_int;
    fn smhv_get_debug_view(ctx: *mut c_void, which: c_int, rgba: *mut u8, w: *mut u32, h: *mut u32) -> c_int;
}

extern "C" fn log_sink(level: c_int, msg: *const c_char) {
    let msg = unsafe { CStr::from_ptr(msg) }.to_string_lossy();
    match level { 1 => log::error!("{msg}"), 2 => log::warn!("{msg}"), 3 => log::info!("{msg}"), _ => log::debug!("{msg}") }
}

fn check(rc: c_int) -> Result<(), AnyError> {
    if rc == 0 { Ok(()) } else {
        Err(anyhow::anyhow!("smh_vision_hip error {rc}: {}", unsafe { CStr::from_ptr(smhv_last_error()) }.to_string_lossy()))
    }
}

pub struct HipInstance {
    ctx: *mut c_void,
    cpu_frame: Arc<VisionFrame>,
    scales: SusRefCell<image::GrayImage>,        // what find_scales_preprocess hands back by pointer
}
unsafe impl Send for HipInstance {}
unsafe impl Sync for HipInstance {}
impl Drop for HipInstance { fn drop(&mut self) { unsafe { smhv_shutdown(self.ctx) } } }

impl Vision for HipInstance {
    type LSDImage = ();                              // opaque to the host, as transmuters::GpuImage is (dylib.rs:157-178)
    type Error = AnyError;

    fn init() -> Result<Self, AnyError> {
        let mut ctx = ptr::null_mut();
        check(unsafe { smhv_init(0, Some(log_sink), &mut ctx) })?;    // Err => hardware.rs falls back to CPUFallback
        Ok(Self { ctx, cpu_frame: Default::default(), scales: Default::default() })
    }
    fn thread_ctx(&self) -> Result<(), AnyError> { check(unsafe { smhv_thread_ctx(self.ctx) }) }
    fn get_cpu_frame(&self) -> Arc<VisionFrame> { self.cpu_frame.clone() }

    fn load_frame(&mut self, image: VisionFrame) -> Result<(), AnyError> {
        // the same two cases as vision-gpu/src/lib.rs:175-179: a view of a larger image is uploaded with a pitched copy
        let (x, y, w, h) = image.bounds();
        let (pw, ph) = image.inner().dimensions();
        if image.inner().bounds() != image.bounds() {
            check(unsafe { smhv_load_frame_view(self.ctx, image.inner().as_ptr(), pw, ph, x, y, w, h) })?;
        } else {
            check(unsafe { smhv_load_frame(self.ctx, image.inner().as_ptr(), w, h) })?;
        }
        self.cpu_frame = Arc::new(image);
        Ok(())
    }

    fn crop_to_map(&self, grayscale: bool) -> Result<Option<(image::RgbaImage, [u32; 4])>, AnyError> {
        let mut roi = [0u32; 4];
        check(unsafe { smhv_map_bounds(self.cpu_frame.width(), self.cpu_frame.height(), roi.as_mut_ptr()) })?;
        let mut ui = vec![0u8; roi[2] as usize * roi[3] as usize * 4];
        let mut open = 0;
        check(unsafe { smhv_crop_to_map(self.ctx, grayscale as c_int, &mut open, roi.as_mut_ptr(), ui.as_mut_ptr()) })?;
        if open == 0 { return Ok(None); }
        Ok(Some((image::RgbaImage::from_vec(roi[2], roi[3], ui).unwrap(), roi)))
    }

    fn ocr_preprocess(&self) -> Result<(*const u8, usize), AnyError> {
        let (mut p, mut n) = (ptr::null(), 0usize);
        check(unsafe { smhv_ocr_preprocess(self.ctx, &mut p, &mut n) })?;
        Ok((p, n))
    }

    fn find_scales_preprocess(&self, scales_start_y: u32) -> Result<*const SusRefCell<image::GrayImage>, AnyError> {
        let (mut p, mut w, mut h) = (ptr::null(), 0u32, 0u32);
        check(unsafe { smhv_find_scales_preprocess(self.ctx, scales_start_y, &mut p, &mut w, &mut h) })?;
        let bytes = unsafe { std::slice::from_raw_parts(p, w as usize * h as usize) }.to_vec();
        *self.scales.borrow_mut() = image::GrayImage::from_vec(w, h, bytes).unwrap();
        Ok(&self.scales as *const _)
    }

    fn isolate_map_markers(&self) -> Result<(), AnyError> { check(unsafe { smhv_isolate_map_markers(self.ctx) }) }
    fn mask_marker_lines(&self) -> Result<(), AnyError> { check(unsafe { smhv_mask_marker_lines(self.ctx) }) }

    fn find_longest_line(&self, _image: &(), pt: Point<f32>, max_gap: f32) -> Result<(Line<f32>, f32), AnyError> {
        let (mut l, mut len) = (SmhvLine::default(), 0f32);
        check(unsafe { smhv_find_longest_line(self.ctx, pt.x, pt.y, max_gap, &mut l, &mut len) })?;
        Ok((Line::new(Point::new(l.x0, l.y0), Point::new(l.x1, l.y1)), len))
    }

    fn find_marker_lines(&self, max_gap: u32) -> Result<SmallVec<Line<f32>, 32>, AnyError> {
        let (mut out, mut n) = ([SmhvLine::default(); 32], 0u32);
        check(unsafe { smhv_find_marker_lines(self.ctx, max_gap, out.as_mut_ptr(), &mut n) })?;
        let mut v = SmallVec::new();
        for l in &out[..n as usize] { v.push(Line::new(Point::new(l.x0, l.y0), Point::new(l.x1, l.y1))); }
        Ok(v)
    }

    fn get_debug_view(&self, choice: debug::DebugView) -> Option<Arc<image::RgbaImage>> {
        if choice == debug::DebugView::None { return None; }
        let (mut w, mut h) = (0u32, 0u32);
        unsafe { smhv_get_debug_view(self.ctx, choice as u8 as c_int, ptr::null_mut(), &mut w, &mut h) };
        let mut px = vec![0u8; w as usize * h as usize * 4];
        if unsafe { smhv_get_debug_view(self.ctx, choice as u8 as c_int, px.as_mut_ptr(), &mut w, &mut h) } != 0 { return None; }
        Some(Arc::new(image::RgbaImage::from_vec(w, h, px)?))
    }
}
