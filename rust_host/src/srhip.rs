//! Raw binding of `include/srhip.h` plus a small safe wrapper.  One `extern` line per C
//! declaration; `tests/test_abi.py` keeps the header, the library and the Python table in step,
//! and `tests/test_rust_host.py` checks that every symbol named here is exported.
#![allow(dead_code)]
use std::ffi::CStr;
use std::os::raw::{c_char, c_int, c_void};
use std::ptr;

#[repr(C)]
pub struct SrCtx {
    _private: [u8; 0],
}

pub const SR_OK: c_int = 0;
pub const SR_E_HIP: c_int = -5;
pub const SR_GRAPH_SR_NET: c_int = 0;
pub const SR_GRAPH_BILINEAR: c_int = 1;
pub const SR_GRAPH_DOWNSAMPLE: c_int = 2;
pub const SR_PRECISION_F32: c_int = 0;
pub const SR_PRECISION_SPLIT_F16: c_int = 1;
pub const SR_FACTOR: c_int = 3;
pub const SR_E_COMM: c_int = -9;
pub const SR_E_DOMAIN: c_int = -10;
pub const SR_HALO: c_int = 7;
pub const SR_COMM_ID_BYTES: c_int = 128;

extern "C" {
    pub fn sr_rsr_decode(blob: *const u8, len: usize, out: *mut f32, cap: usize, n_out: *mut usize) -> c_int;
    pub fn sr_rsr_encode(params: *const f32, n: usize, out: *mut u8, cap: usize, len_out: *mut usize) -> c_int;
    pub fn sr_create(out: *mut *mut SrCtx, params: *const f32, n_params: usize, factor: c_int, device: c_int) -> c_int;
    pub fn sr_create_graph(out: *mut *mut SrCtx, graph: c_int, params: *const f32, n_params: usize, factor: c_int,
                           device: c_int) -> c_int;
    pub fn sr_num_params(graph: c_int) -> c_int;
    pub fn sr_num_params_factor(factor: c_int) -> c_int;
    pub fn sr_set_precision(ctx: *mut SrCtx, mode: c_int) -> c_int;
    pub fn sr_check_domain(ctx: *mut SrCtx) -> c_int;
    pub fn sr_destroy(ctx: *mut SrCtx);
    pub fn sr_upscale_f32(ctx: *mut SrCtx, input: *const f32, n: c_int, h: c_int, w: c_int, out: *mut f32) -> c_int;
    pub fn sr_upscale_rgba8(ctx: *mut SrCtx, input: *const u8, in_channels: c_int, n: c_int, h: c_int, w: c_int,
                            out_rgba: *mut u8) -> c_int;
    pub fn sr_reserve_f32(ctx: *mut SrCtx, n: c_int, h: c_int, w: c_int) -> c_int;
    pub fn sr_reserve_rgba8(ctx: *mut SrCtx, in_channels: c_int, n: c_int, h: c_int, w: c_int) -> c_int;
    pub fn sr_upscale_f32_dev(ctx: *mut SrCtx, d_in: *const f32, n: c_int, h: c_int, w: c_int, d_out: *mut f32,
                              stream: *mut c_void) -> c_int;
    pub fn sr_upscale_rgba8_dev(ctx: *mut SrCtx, d_in: *const u8, in_channels: c_int, n: c_int, h: c_int, w: c_int,
                                d_out: *mut u8, stream: *mut c_void) -> c_int;
    pub fn sr_upscale_band_f32_dev(ctx: *mut SrCtx, d_in: *const f32, h_ext: c_int, w: c_int, halo_top: c_int,
                                   halo_bot: c_int, d_out: *mut f32, stream: *mut c_void) -> c_int;
    pub fn sr_upscale_band_rgba8_dev(ctx: *mut SrCtx, d_in: *const u8, in_channels: c_int, h_ext: c_int, w: c_int,
                                     halo_top: c_int, halo_bot: c_int, d_out: *mut u8, stream: *mut c_void) -> c_int;
    pub fn sr_read_feature(ctx: *mut SrCtx, which: c_int, out_host: *mut f32, cap_floats: usize) -> c_int;
    pub fn sr_upscale_f32_multi(ctxs: *const *mut SrCtx, n_ctx: c_int, input: *const f32, h: c_int, w: c_int, out: *mut f32) -> c_int;
    pub fn sr_upscale_rgba8_multi(ctxs: *const *mut SrCtx, n_ctx: c_int, input: *const u8, in_channels: c_int, h: c_int, w: c_int,
                                  out_rgba: *mut u8) -> c_int;
    pub fn sr_upscale_f32_batch_multi(ctxs: *const *mut SrCtx, n_ctx: c_int, input: *const f32, n: c_int, h: c_int, w: c_int,
                                      out: *mut f32) -> c_int;
    pub fn sr_upscale_rgba8_batch_multi(ctxs: *const *mut SrCtx, n_ctx: c_int, input: *const u8, in_channels: c_int, n: c_int,
                                        h: c_int, w: c_int, out_rgba: *mut u8) -> c_int;
    // RCCL communicator inside the library + device-resident row bands (one image over several GPUs)
    pub fn sr_comm_available() -> c_int;
    pub fn sr_comm_unique_id(id: *mut u8, cap: usize) -> c_int;
    pub fn sr_comm_init_rank(ctx: *mut SrCtx, id: *const u8, id_len: usize, rank: c_int, nranks: c_int) -> c_int;
    pub fn sr_comm_init_all(ctxs: *const *mut SrCtx, n: c_int) -> c_int;
    pub fn sr_comm_init_local(ctxs: *const *mut SrCtx, n: c_int) -> c_int;
    pub fn sr_comm_destroy(ctx: *mut SrCtx);
    pub fn sr_comm_rank(ctx: *mut SrCtx, rank: *mut c_int, nranks: *mut c_int) -> c_int;
    pub fn sr_last_comm_error(ctx: *mut SrCtx) -> c_int;
    pub fn sr_last_comm_ms(ctx: *mut SrCtx, comm_ms: *mut f64) -> c_int;
    pub fn sr_last_comm_exposed_ms(ctx: *mut SrCtx, exposed_ms: *mut f64) -> c_int;
    pub fn sr_upscale_sharded_f32_dev(ctx: *mut SrCtx, d_band: *const f32, h_band: c_int, w: c_int, d_out: *mut f32,
                                      stream: *mut c_void) -> c_int;
    pub fn sr_upscale_sharded_rgba8_dev(ctx: *mut SrCtx, d_band: *const u8, in_channels: c_int, h_band: c_int, w: c_int,
                                        d_out: *mut u8, stream: *mut c_void) -> c_int;
    pub fn sr_upscale_sharded_f32_all(ctxs: *const *mut SrCtx, n: c_int, d_bands: *const *const f32, h_bands: *const c_int,
                                      w: c_int, d_outs: *const *mut f32) -> c_int;
    pub fn sr_upscale_sharded_rgba8_all(ctxs: *const *mut SrCtx, n: c_int, d_bands: *const *const u8, in_channels: c_int,
                                        h_bands: *const c_int, w: c_int, d_outs: *const *mut u8) -> c_int;
    pub fn sr_set_pipeline(ctx: *mut SrCtx, enabled: c_int) -> c_int;
    pub fn sr_host_alloc(out: *mut *mut c_void, bytes: usize) -> c_int;  // page-locked host memory
    pub fn sr_host_free(p: *mut c_void);
    pub fn sr_set_profiling(ctx: *mut SrCtx, enabled: c_int) -> c_int;
    pub fn sr_last_timing(ctx: *mut SrCtx, total_ms: *mut f64, stage_ms: *mut f64, h2d_ms: *mut f64, d2h_ms: *mut f64) -> c_int;
    pub fn sr_device_info(ctx: *mut SrCtx, name: *mut c_char, cap: usize, cus: *mut c_int, mhz: *mut c_int) -> c_int;
    pub fn sr_last_hip_error(ctx: *mut SrCtx) -> c_int;
    pub fn sr_strerror(status: c_int) -> *const c_char;
}

/// Text of an `sr_status`; for SR_E_PARAM_COUNT / SR_E_BYTEVEC it is the reference's own panic text.
pub fn strerror(rc: c_int) -> String {
    unsafe { CStr::from_ptr(sr_strerror(rc)) }.to_string_lossy().into_owned()
}

/// `<Vec<f32>>::decode::<u32>(&bytes)` of the reference, through the library's parser.
pub fn rsr_decode(blob: &[u8]) -> Result<Vec<f32>, String> {
    let mut n = 0usize;
    let rc = unsafe { sr_rsr_decode(blob.as_ptr(), blob.len(), ptr::null_mut(), 0, &mut n) };
    if rc != SR_OK {
        return Err(strerror(rc));
    }
    let mut out = vec![0f32; n];
    let rc = unsafe { sr_rsr_decode(blob.as_ptr(), blob.len(), out.as_mut_ptr(), n, &mut n) };
    if rc != SR_OK {
        return Err(strerror(rc));
    }
    Ok(out)
}

/// 128 opaque bytes of ncclGetUniqueId for `Engine::comm_init_rank` (rank 0 only).
pub fn comm_unique_id() -> Result<Vec<u8>, String> {
    let mut id = vec![0u8; SR_COMM_ID_BYTES as usize];
    let rc = unsafe { sr_comm_unique_id(id.as_mut_ptr(), id.len()) };
    if rc == SR_OK { Ok(id) } else { Err(strerror(rc)) }
}

/// Owning handle of one engine context: a graph, its parameters, one GPU.
pub struct Engine {
    ctx: *mut SrCtx,
    graph: c_int,
}

impl Engine {
    pub fn new(graph: c_int, params: &[f32], device: c_int) -> Result<Engine, String> {
        let mut ctx = ptr::null_mut();
        let p = if params.is_empty() { ptr::null() } else { params.as_ptr() };
        let rc = unsafe { sr_create_graph(&mut ctx, graph, p, params.len(), SR_FACTOR, device) };
        if rc != SR_OK {
            return Err(strerror(rc));
        }
        Ok(Engine { ctx: ctx, graph: graph })
    }

    pub fn set_precision(&mut self, mode: c_int) -> Result<(), String> {
        let rc = unsafe { sr_set_precision(self.ctx, mode) };
        if rc == SR_OK { Ok(()) } else { Err(strerror(rc)) }
    }

    /// Output (width, height) for an input of (w, h): x3, or /3 for the downsample graph.
    pub fn out_dims(&self, w: u32, h: u32) -> (u32, u32) {
        if self.graph == SR_GRAPH_DOWNSAMPLE { (w / 3, h / 3) } else { (w * 3, h * 3) }
    }

    /// img_to_data + graph.forward + data_to_img(..).to_rgba() in one device pass.
    pub fn upscale_rgba8(&mut self, rgba: &[u8], w: u32, h: u32) -> Result<Vec<u8>, String> {
        assert_eq!(rgba.len(), w as usize * h as usize * 4);
        let (ow, oh) = self.out_dims(w, h);
        let mut out = vec![0u8; ow as usize * oh as usize * 4];
        let rc = unsafe { sr_upscale_rgba8(self.ctx, rgba.as_ptr(), 4, 1, h as c_int, w as c_int, out.as_mut_ptr()) };
        if rc == SR_OK {
            Ok(out)
        } else if rc == SR_E_HIP {
            Err(format!("{} (hipError {})", strerror(rc), unsafe { sr_last_hip_error(self.ctx) }))
        } else {
            Err(strerror(rc))
        }
    }

    /// Same seam as the reference's `graph.forward(1, vec![input], &params)`: NHWC f32 in and out.
    pub fn forward_f32(&mut self, values: &[f32], w: u32, h: u32) -> Result<Vec<f32>, String> {
        assert_eq!(values.len(), w as usize * h as usize * 3);
        let (ow, oh) = self.out_dims(w, h);
        let mut out = vec![0f32; ow as usize * oh as usize * 3];
        let rc = unsafe { sr_upscale_f32(self.ctx, values.as_ptr(), 1, h as c_int, w as c_int, out.as_mut_ptr()) };
        if rc == SR_OK { Ok(out) } else { Err(strerror(rc)) }
    }

    /// One image over several GPUs of this process: image (w x h, RGBA) split into `engines.len()` row shares, one per
    /// engine / device, halo rows read from `rgba` itself (sr_upscale_rgba8_multi).
    pub fn upscale_rgba8_multi(engines: &mut [Engine], rgba: &[u8], w: u32, h: u32) -> Result<Vec<u8>, String> {
        assert_eq!(rgba.len(), w as usize * h as usize * 4);
        let ctxs: Vec<*mut SrCtx> = engines.iter().map(|e| e.ctx).collect();
        let mut out = vec![0u8; w as usize * 3 * h as usize * 3 * 4];
        let rc = unsafe {
            sr_upscale_rgba8_multi(ctxs.as_ptr(), ctxs.len() as c_int, rgba.as_ptr(), 4, h as c_int, w as c_int, out.as_mut_ptr())
        };
        if rc == SR_OK { Ok(out) } else { Err(strerror(rc)) }
    }

    /// A batch dealt round-robin (image i -> engine i mod N), one host thread per engine inside the library.
    pub fn upscale_rgba8_batch_multi(engines: &mut [Engine], rgba: &[u8], n: u32, w: u32, h: u32) -> Result<Vec<u8>, String> {
        assert_eq!(rgba.len(), n as usize * w as usize * h as usize * 4);
        let ctxs: Vec<*mut SrCtx> = engines.iter().map(|e| e.ctx).collect();
        let mut out = vec![0u8; n as usize * w as usize * 3 * h as usize * 3 * 4];
        let rc = unsafe {
            sr_upscale_rgba8_batch_multi(ctxs.as_ptr(), ctxs.len() as c_int, rgba.as_ptr(), 4, n as c_int, h as c_int, w as c_int,
                                         out.as_mut_ptr())
        };
        if rc == SR_OK { Ok(out) } else { Err(strerror(rc)) }
    }

    /// One process per GPU: join the band communicator.  Rank 0 obtains `id` from `comm_unique_id()` and hands it to
    /// the other ranks (file, socket, environment) before every rank calls this.
    pub fn comm_init_rank(&mut self, id: &[u8], rank: c_int, nranks: c_int) -> Result<(), String> {
        let rc = unsafe { sr_comm_init_rank(self.ctx, id.as_ptr(), id.len(), rank, nranks) };
        if rc == SR_OK { Ok(()) } else { Err(format!("{} (ncclResult {})", strerror(rc), unsafe { sr_last_comm_error(self.ctx) })) }
    }

    pub fn last_timing(&mut self) -> (f64, f64, f64) {
        let (mut total, mut h2d, mut d2h) = (0f64, 0f64, 0f64);
        unsafe { sr_last_timing(self.ctx, &mut total, ptr::null_mut(), &mut h2d, &mut d2h) };
        (total, h2d, d2h)
    }
}

impl Drop for Engine {
    fn drop(&mut self) {
        unsafe { sr_destroy(self.ctx) }
    }
}
