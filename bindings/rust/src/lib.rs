//! `aprilgrid::detector::TagDetector` over libaprilgrid_amd.so (MI355X / gfx950).
//!
//! The surface is the reference's (aprilgrid 0.8.0): `TagDetector::new(&TagFamily, Option<DetectorParams>)`
//! (src/detector.rs:364-406), `detect(&self, &DynamicImage) -> HashMap<u32, [(f32, f32); 4]>` (:505-540),
//! `refined_saddle_points(&self, &DynamicImage) -> Vec<Saddle>` (:408-446), `detect_kornia` behind the `kornia`
//! feature (:478-503), plus `detect_many` (no counterpart there: `detect` over a batch of equally sized frames, the
//! form that keeps a GPU busy).  The types `TagFamily`, `DetectorParams` and `Saddle` repeat the reference's
//! definitions (src/tag_families.rs:5-28, src/detector.rs:25-41, src/saddle.rs:3-15) so that a caller switches by
//! changing a `use` line; inside the reference crate the same file works as `src/amd.rs` with these three replaced by
//! `crate::` paths.
//!
//! Failure model: the reference returns no errors -- an internal failure is a panic (src/detector.rs:500), "nothing
//! found" an empty collection.  Every C entry point returns a status and never unwinds into Rust; a non-zero status
//! becomes the panic the reference's own failures are.
//!
//! Not compiled in the repository's build image (no Rust toolchain there).  The declarations of `ffi` are checked
//! against the header by the repository's CPU test-suite (tests/test_rust_binding.py).
pub mod ffi;

use std::collections::HashMap;
use std::os::raw::{c_int, c_void};
use std::sync::Mutex;

/// reference src/tag_families.rs:5-13 -- the discriminants are `agx_family` (checked by tests/test_rust_binding.py).
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
#[repr(i32)]
pub enum TagFamily {
    T16H5 = 0,
    T25H7 = 1,
    T25H9 = 2,
    T36H11 = 3,
    /// 1 bit Border
    T36H11B1 = 4,
}

impl std::str::FromStr for TagFamily {
    type Err = std::fmt::Error;
    /// reference src/tag_families.rs:15-28, through `agx_family_from_str` (one table, the library's).
    fn from_str(s: &str) -> Result<Self, Self::Err> {
        let name = std::ffi::CString::new(s).map_err(|_| std::fmt::Error)?;
        let mut fam: c_int = -1;
        let st = unsafe { ffi::agx_family_from_str(name.as_ptr(), &mut fam) };
        match (st, fam) {
            (ffi::AGX_OK, ffi::AGX_T16H5) => Ok(TagFamily::T16H5),
            (ffi::AGX_OK, ffi::AGX_T25H7) => Ok(TagFamily::T25H7),
            (ffi::AGX_OK, ffi::AGX_T25H9) => Ok(TagFamily::T25H9),
            (ffi::AGX_OK, ffi::AGX_T36H11) => Ok(TagFamily::T36H11),
            (ffi::AGX_OK, ffi::AGX_T36H11B1) => Ok(TagFamily::T36H11B1),
            _ => Err(std::fmt::Error),
        }
    }
}

/// reference src/detector.rs:25-41
#[derive(Debug, Clone, Copy)]
pub struct DetectorParams {
    pub tag_spacing_ratio: f32,
    pub min_saddle_angle: f32,
    pub max_saddle_angle: f32,
    pub max_num_of_boards: u8,
}

impl DetectorParams {
    pub fn default_params() -> DetectorParams {
        DetectorParams { tag_spacing_ratio: 0.3, min_saddle_angle: 30.0, max_saddle_angle: 60.0, max_num_of_boards: 2 }
    }
}

/// reference src/saddle.rs:3-15
#[derive(Debug, Clone, Copy)]
pub struct Saddle {
    pub p: (f32, f32),
    pub k: f32,
    pub theta: f32,
    pub phi: f32,
}

impl Saddle {
    pub const fn arr(&self) -> [f32; 2] {
        [self.p.0, self.p.1]
    }
}

/// What the ABI is handed for one image: borrowed pixels of a natively supported variant, or the two planes the
/// reference derives itself (src/detector.rs:409 `to_luma32f`, :507 `to_luma8`) for every other variant.
enum Input<'a> {
    Native { px: *const c_void, stride: usize, fmt: c_int, _keep: std::marker::PhantomData<&'a ()> },
    Planes { luma32f: image::ImageBuffer<image::Luma<f32>, Vec<f32>>, luma8: image::GrayImage },
}

/// A pooled handle; back in the pool when dropped -- also when the closure that used it panics (a handle owns about
/// 5 GB of device workspace at 256 frames of 1280 x 800 and must not leak).
struct Lease<'a> {
    pool: &'a Mutex<Vec<Handle>>,
    h: Option<Handle>,
}
impl Drop for Lease<'_> {
    fn drop(&mut self) {
        if let Some(h) = self.h.take() {
            match self.pool.lock() {
                Ok(mut p) => p.push(h),
                Err(_) => unsafe { ffi::agx_detector_destroy(h.0) },
            }
        }
    }
}

/// `*mut agx_detector` that may move between threads (a handle is one device + stream: used by one thread at a time,
/// which the pool guarantees).
struct Handle(*mut ffi::agx_detector);
unsafe impl Send for Handle {}

/// Same surface as `aprilgrid::detector::TagDetector`.  The reference's detector is `Send + Sync` and `detect(&self)`
/// may be called from any number of threads (src/detector.rs:17-23); here every call leases a handle from a pool
/// (created on demand), so the property holds.
pub struct TagDetector {
    pool: Mutex<Vec<Handle>>,
    family: c_int,
    params: ffi::agx_params,
    device: c_int,
}

impl TagDetector {
    /// reference src/detector.rs:364-406.  Infallible like the reference's: the device is touched by the first call.
    pub fn new(tag_family: &TagFamily, optional_detector_params: Option<DetectorParams>) -> TagDetector {
        Self::new_on_device(tag_family, optional_detector_params, 0)
    }

    /// The same on GPU `device` of the node (one `TagDetector` per GPU shards a stream of frames; frames are independent).
    pub fn new_on_device(tag_family: &TagFamily, optional_detector_params: Option<DetectorParams>, device: i32) -> TagDetector {
        let p = optional_detector_params.unwrap_or(DetectorParams::default_params());
        TagDetector {
            pool: Mutex::new(Vec::new()),
            family: *tag_family as c_int,
            device: device as c_int,
            params: ffi::agx_params {
                tag_spacing_ratio: p.tag_spacing_ratio,
                min_saddle_angle: p.min_saddle_angle,
                max_saddle_angle: p.max_saddle_angle,
                max_num_of_boards: p.max_num_of_boards,
            },
        }
    }

    fn with_handle<R>(&self, f: impl FnOnce(*mut ffi::agx_detector) -> R) -> R {
        let pooled = self.pool.lock().unwrap().pop();
        let h = pooled.unwrap_or_else(|| {
            let mut h: *mut ffi::agx_detector = std::ptr::null_mut();
            let st = unsafe { ffi::agx_detector_create(self.family, &self.params, self.device, &mut h) };
            assert_eq!(st, ffi::AGX_OK, "agx_detector_create failed: {} ({})", st, last_error(std::ptr::null()));
            Handle(h)
        });
        let lease = Lease { pool: &self.pool, h: Some(h) };
        f(lease.h.as_ref().unwrap().0)
    }

    fn input(img: &image::DynamicImage) -> (Input<'_>, u32, u32) {
        use image::DynamicImage::*;
        let native = |px: *const c_void, stride: usize, fmt: c_int| Input::Native { px, stride, fmt, _keep: std::marker::PhantomData };
        match img {
            // the three variants the reference's tests, benches and detect_kornia feed the path: no copy, the luma
            // conversion is fused into the blur kernel's load stage
            ImageLuma8(b) => (native(b.as_raw().as_ptr() as *const c_void, b.width() as usize, ffi::AGX_L8), b.width(), b.height()),
            ImageLuma16(b) => (native(b.as_raw().as_ptr() as *const c_void, 2 * b.width() as usize, ffi::AGX_L16), b.width(), b.height()),
            ImageRgb8(b) => (native(b.as_raw().as_ptr() as *const c_void, 3 * b.width() as usize, ffi::AGX_RGB8), b.width(), b.height()),
            // La8, Rgba8, Rgb16, Rgba16, La16, Rgb32F, Rgba32F: exactly the planes the reference computes
            other => (Input::Planes { luma32f: other.to_luma32f(), luma8: other.to_luma8() }, other.width(), other.height()),
        }
    }

    /// reference src/detector.rs:408-446 -- the hot path.
    pub fn refined_saddle_points(&self, img: &image::DynamicImage) -> Vec<Saddle> {
        let (inp, w, h) = Self::input(img);
        let (px, stride, fmt) = match &inp {
            Input::Native { px, stride, fmt, .. } => (*px, *stride, *fmt),
            Input::Planes { luma32f, .. } => (luma32f.as_raw().as_ptr() as *const c_void, 4 * w as usize, ffi::AGX_LF32),
        };
        let mut out = vec![ffi::agx_saddle::default(); 4096];
        let mut n = 0u32;
        let call = |out: &mut Vec<ffi::agx_saddle>, n: &mut u32| {
            self.with_handle(|d| unsafe {
                ffi::agx_refined_saddle_points(d, px, w as c_int, h as c_int, stride, fmt, out.as_mut_ptr(), out.len() as u32, n)
            })
        };
        let mut st = call(&mut out, &mut n);
        if st == ffi::AGX_ERR_CAPACITY && n as usize > out.len() {
            // Vec<Saddle> has no limit: the call reported the size it needs
            out.resize(n as usize, ffi::agx_saddle::default());
            st = call(&mut out, &mut n);
        }
        assert_eq!(st, ffi::AGX_OK, "agx_refined_saddle_points failed: {}", st);
        out[..n as usize].iter().map(|s| Saddle { p: (s.x, s.y), k: s.k, theta: s.theta, phi: s.phi }).collect()
    }

    fn tags_to_map(out: &[ffi::agx_tag]) -> HashMap<u32, [(f32, f32); 4]> {
        // later entries replace earlier ones, as HashMap::insert does at src/detector.rs:520
        out.iter().map(|t| (t.id, [(t.xy[0], t.xy[1]), (t.xy[2], t.xy[3]), (t.xy[4], t.xy[5]), (t.xy[6], t.xy[7])])).collect()
    }

    /// reference src/detector.rs:505-540
    pub fn detect(&self, img: &image::DynamicImage) -> HashMap<u32, [(f32, f32); 4]> {
        let (inp, w, h) = Self::input(img);
        let mut out = vec![ffi::agx_tag { id: 0, xy: [0.0; 8] }; 1024];
        let mut n = 0u32;
        let st = self.with_handle(|d| unsafe {
            match &inp {
                Input::Native { px, stride, fmt, .. } => {
                    ffi::agx_detect(d, *px, w as c_int, h as c_int, *stride, *fmt, out.as_mut_ptr(), out.len() as u32, &mut n)
                }
                Input::Planes { luma32f, luma8 } => ffi::agx_detect_planes(
                    d, luma32f.as_raw().as_ptr(), 4 * w as usize, luma8.as_raw().as_ptr(), w as usize, w as c_int, h as c_int,
                    out.as_mut_ptr(), out.len() as u32, &mut n,
                ),
            }
        });
        assert_eq!(st, ffi::AGX_OK, "agx_detect failed: {}", st);
        Self::tags_to_map(&out[..n as usize])
    }

    /// reference src/detector.rs:478-503: `kornia::image::Image<u8, N>`, N = 1 (u8c1) or 3 (u8c3, HWC interleaved); any
    /// other N panics with the reference's message.  The tensor's storage is handed over as it is (the reference
    /// clones it into a GrayImage / RgbImage first).
    #[cfg(feature = "kornia")]
    pub fn detect_kornia<const N: usize>(&self, img: &kornia::image::Image<u8, N>) -> HashMap<u32, [(f32, f32); 4]> {
        let fmt = match img.num_channels() {
            1 => ffi::AGX_L8,
            3 => ffi::AGX_RGB8,
            _ => panic!("Only support u8c1 and u8c3"),
        };
        let (w, h) = (img.width(), img.height());
        let px: &[u8] = img.as_slice(); // H x W x N, row-major, tightly packed
        debug_assert_eq!(px.len(), w * h * N);
        let mut out = vec![ffi::agx_tag { id: 0, xy: [0.0; 8] }; 1024];
        let mut n = 0u32;
        let st = self.with_handle(|d| unsafe {
            ffi::agx_detect(d, px.as_ptr() as *const c_void, w as c_int, h as c_int, N * w, fmt, out.as_mut_ptr(), out.len() as u32, &mut n)
        });
        assert_eq!(st, ffi::AGX_OK, "agx_detect failed: {}", st);
        Self::tags_to_map(&out[..n as usize])
    }

    /// `detect` of every frame of a tightly packed `[n][h][w]` u8 buffer (no counterpart in the reference).  Upload,
    /// saddle chain, board search and decode run chunk by chunk on the device (`agx_detect_batch`; frames the device
    /// cannot decide take the same search on a pool of host threads, 0 = every CPU the process is granted).  The maps
    /// are those of calling `detect` per frame.
    pub fn detect_many(&self, frames: &[u8], n: usize, w: u32, h: u32) -> Vec<HashMap<u32, [(f32, f32); 4]>> {
        assert_eq!(frames.len(), n * (w as usize) * (h as usize));
        const CAP: usize = 256;
        let mut out = vec![ffi::agx_tag { id: 0, xy: [0.0; 8] }; n * CAP];
        let mut counts = vec![0u32; n];
        let mut status = vec![0 as c_int; n];
        let st = self.with_handle(|d| unsafe {
            ffi::agx_detect_batch(
                d, frames.as_ptr() as *const c_void, std::ptr::null(), n as c_int, w as c_int, h as c_int, w as usize,
                (w as usize) * (h as usize), ffi::AGX_L8, out.as_mut_ptr(), CAP as u32, counts.as_mut_ptr(), status.as_mut_ptr(), 0,
            )
        });
        assert_eq!(st, ffi::AGX_OK, "agx_detect_batch failed: {}", st);
        (0..n).map(|i| Self::tags_to_map(&out[i * CAP..i * CAP + counts[i] as usize])).collect()
    }
}

impl Drop for TagDetector {
    fn drop(&mut self) {
        let handles = match self.pool.lock() {
            Ok(mut p) => std::mem::take(&mut *p),
            Err(e) => std::mem::take(&mut *e.into_inner()),
        };
        for h in handles {
            unsafe { ffi::agx_detector_destroy(h.0) }
        }
    }
}

fn last_error(det: *const ffi::agx_detector) -> String {
    unsafe {
        let p = ffi::agx_last_error(det);
        if p.is_null() { String::new() } else { std::ffi::CStr::from_ptr(p).to_string_lossy().into_owned() }
    }
}
