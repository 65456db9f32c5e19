//! `sdf-gpu`: the MI355X path of the `sdf` crate behind its own types.
//!
//! Declarations follow `include/sdfgpu.h` one to one; record layouts are the crate's own
//! `#[repr(C)]` structs (`component.rs:11-78`), checked below against the sizes
//! `include/sdf_blocks.h` asserts.  This file is documentation that happens to be source:
//! the build image has no Rust toolchain, so it has not been compiled here.

use bevy::math::{Vec3, Vec4};
use sdf::component::{SdfBoundingBoxBlock, SdfOperationBlock, SdfOperationUptreeBlock, SdfTreeBuffer};
use std::ffi::CStr;
use std::os::raw::{c_char, c_int, c_void};

// glam with SSE2 (bevy's default on x86-64); a `scalar-math` build has another layout
const _: () = assert!(std::mem::size_of::<SdfOperationBlock>() == 624);
const _: () = assert!(std::mem::size_of::<SdfOperationUptreeBlock>() == 224);
const _: () = assert!(std::mem::size_of::<SdfBoundingBoxBlock>() == 208);

#[repr(C)]
pub struct Ctx {
    _private: [u8; 0],
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct Camera {
    pub eye: [f32; 3],
    pub origin: [f32; 3],
    pub du: [f32; 3],
    pub dv: [f32; 3],
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct MarchParams {
    pub max_steps: u32,
    pub eps_hit: f32,
    pub t_min: f32,
    pub t_max: f32,
    pub normal_h: f32,
    pub flags: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct Stats {   // sdfgpu_stats
    pub queries: u64, pub inner_visits: u64, pub sphere_evals: u64, pub box_evals: u64,
    pub wave_inner: u64, pub wave_leaf: u64, pub steps: u64, pub rays: u64,
    pub wave_steps: u64, pub wave_walks: u64, pub inner_needed: u64, pub sphere_needed: u64, pub box_needed: u64,
    pub list_steps: u64, pub wave_lists: u64, pub wave_list_evals: u64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct TreeInfo {
    pub n_nodes: u32,
    pub n_prims: u32,
    pub n_inner: u32,
    pub max_level: u32,
    pub bounds: u32,
    pub identity_inner: u32,
    pub rigid_leaves: u32,
    pub stack_depth: u32,
    pub device_bytes: u64,
    pub scene_lo: [f32; 3],
    pub scene_hi: [f32; 3],
    /// `sdfgpu_extension` bits in effect in this tree (0 = evaluated as the crate is written)
    pub extensions: u32,
    /// cells of the nearest-candidate grid built for the march (0 = none)
    pub grid_cells: u32,
    /// sdfgpu_walk: 0 hierarchy, 1 chain, 2 general pruned, 3 general exhaustive
    pub walk: u32,
    pub reserved: u32,
}

pub const LAYOUT_GLAM_SSE2: u32 = 0;
pub const BOUNDS_AUTO: u32 = 0;
pub const NO_PRIM: u32 = 0xFFFF_FFFF;
/// `sdfgpu_option` (none of them changes a result)
pub const OPT_REORDER: i32 = 0;
pub const OPT_SKIP_EXTENSION: i32 = 1;
pub const OPT_COMPACTION: i32 = 2;
pub const OPT_TILE_ORDER: i32 = 3;
pub const OPT_STATS_ASM_WALK: i32 = 4;
pub const OPT_QUERY_PIPELINE: i32 = 5;
pub const OPT_CANDIDATE_GRID: i32 = 6;
/// `sdfgpu_extension`: opt-in evaluation of data the crate carries but never evaluates
/// (elements.rs:207-211 smooth_radius, elements.rs:120-152 SdfBoxFrame, faux_shader.rs:90 clone).
pub const EXT_SMOOTH_UNION: u32 = 1;
pub const EXT_BOX_FRAME: u32 = 2;
pub const EXT_CLONE_REPAIR: u32 = 4;
/// sdfgpu_walk (SdfGpuTreeInfo::walk): which resident form serves the tree; results never depend on it
pub const WALK_HIERARCHY: u32 = 0;
pub const WALK_CHAIN: u32 = 1;
pub const WALK_GENERAL_PRUNED: u32 = 2;
pub const WALK_GENERAL_EXHAUSTIVE: u32 = 3;
/// `sdfgpu_nearest_dev_static` takes at most this many points per call.
pub const STATIC_MAX_POINTS: usize = 262_144;
pub const COMM_ID_BYTES: usize = 128;

/// Opaque `sdfgpu_comm` (an RCCL communicator of the library, one rank per process).
#[repr(C)]
pub struct Comm {
    _private: [u8; 0],
}

extern "C" {
    fn sdfgpu_create(device_id: c_int, out: *mut *mut Ctx) -> c_int;
    fn sdfgpu_destroy(ctx: *mut Ctx);
    fn sdfgpu_last_error(ctx: *const Ctx) -> *const c_char;
    fn sdfgpu_upload_tree(ctx: *mut Ctx, dt: *const SdfOperationBlock, ut: *const SdfOperationUptreeBlock,
                          buffer_len: u32, layout: u32, bounds: u32) -> c_int;
    fn sdfgpu_upload_tree_ext(ctx: *mut Ctx, dt: *const SdfOperationBlock, ut: *const SdfOperationUptreeBlock,
                              buffer_len: u32, layout: u32, bounds: u32, extensions: u32) -> c_int;
    fn sdfgpu_tree_info_get(ctx: *const Ctx, out: *mut TreeInfo) -> c_int;
    pub fn sdfgpu_nearest_dev(ctx: *mut Ctx, d_pts_xyzw: *const f32, n: usize, d_out_dist: *mut f32,
                              d_out_prim: *mut u32, stream: *mut c_void) -> c_int;
    /// allocation-free variant (no reference counterpart): scratch in a static __device__ array
    pub fn sdfgpu_nearest_dev_static(ctx: *mut Ctx, d_pts_xyzw: *const f32, n: usize, d_out_dist: *mut f32,
                                     d_out_prim: *mut u32, stream: *mut c_void) -> c_int;
    pub fn sdfgpu_knn_dev(ctx: *mut Ctx, d_pts_xyzw: *const f32, n: usize, k: u32, d_out_dist: *mut f32,
                          d_out_prim: *mut u32, stream: *mut c_void) -> c_int;
    // multi-GPU framebuffer assembly: one process per GPU, RCCL grouped send/recv to `root`
    pub fn sdfgpu_comm_get_id(ctx: *mut Ctx, id_out: *mut u8) -> c_int;
    pub fn sdfgpu_comm_create(ctx: *mut Ctx, id: *const u8, rank: c_int, world: c_int, out: *mut *mut Comm) -> c_int;
    pub fn sdfgpu_comm_destroy(comm: *mut Comm);
    pub fn sdfgpu_render_frame_comm(ctx: *mut Ctx, comm: *mut Comm, cam: *const Camera, width: u32, height: u32,
                                    mp: *const MarchParams, root: c_int, d_hit: *mut u8, d_depth: *mut f32,
                                    d_normal_xyz: *mut f32, d_prim: *mut u32, stream: *mut c_void) -> c_int;
    // one process driving several GPUs: a clique over n contexts (ncclCommInitAll), every rank's frame share from
    // the one calling thread inside ONE ncclGroupStart / GroupEnd
    pub fn sdfgpu_comm_create_all(ctxs: *const *mut Ctx, n: c_int, comms_out: *mut *mut Comm) -> c_int;
    pub fn sdfgpu_render_frame_all(comms: *const *mut Comm, n: c_int, cam: *const Camera, width: u32, height: u32,
                                   mp: *const MarchParams, root: c_int, d_hit: *mut u8, d_depth: *mut f32,
                                   d_normal_xyz: *mut f32, d_prim: *mut u32, streams: *const *mut c_void) -> c_int;
    // query-batch partitioner: contiguous slices of a host batch over several contexts, results assembled in place
    pub fn sdfgpu_query_slice(n: usize, rank: u32, world: u32, first: *mut usize, count: *mut usize);
    pub fn sdfgpu_nearest_sharded(ctxs: *const *mut Ctx, n_ctx: c_int, pts_xyzw: *const f32, n: usize,
                                  out_dist: *mut f32, out_prim: *mut u32) -> c_int;
    pub fn sdfgpu_knn_sharded(ctxs: *const *mut Ctx, n_ctx: c_int, pts_xyzw: *const f32, n: usize, k: u32,
                              out_dist: *mut f32, out_prim: *mut u32) -> c_int;
    // the pieces of the same assembly for a host that brings its own collective: a rank's tile share into a
    // packed buffer of sdfgpu_packed_share_bytes, every rank's buffer into the frame in one launch
    pub fn sdfgpu_render_tiles_dev(ctx: *mut Ctx, cam: *const Camera, width: u32, height: u32, mp: *const MarchParams,
                                   tile_first: u32, tile_stride: u32, tile_count: u32, d_hit: *mut u8, d_depth: *mut f32,
                                   d_normal_xyz: *mut f32, d_prim: *mut u32, stream: *mut c_void) -> c_int;
    pub fn sdfgpu_packed_share_bytes(width: u32, height: u32, world: u32) -> usize;
    pub fn sdfgpu_untile_gathered_dev(ctx: *mut Ctx, width: u32, height: u32, world: u32, self_rank: u32,
                                      d_gathered: *const c_void, d_mine: *const c_void, d_hit: *mut u8, d_depth: *mut f32,
                                      d_normal_xyz: *mut f32, d_prim: *mut u32, stream: *mut c_void) -> c_int;
    fn sdfgpu_nearest(ctx: *mut Ctx, pts_xyzw: *const f32, n: usize, out_dist: *mut f32, out_prim: *mut u32) -> c_int;
    fn sdfgpu_knn(ctx: *mut Ctx, pts_xyzw: *const f32, n: usize, k: u32, out_dist: *mut f32, out_prim: *mut u32) -> c_int;
    fn sdfgpu_box_bounds(ctx: *mut Ctx, bbox: *const SdfBoundingBoxBlock, pts_xyzw: *const f32, n: usize,
                         out_min: *mut f32, out_max: *mut f32) -> c_int;
    fn sdfgpu_render(ctx: *mut Ctx, cam: *const Camera, width: u32, height: u32, mp: *const MarchParams,
                     hit: *mut u8, depth: *mut f32, normal_xyz: *mut f32, prim: *mut u32) -> c_int;
    fn sdfgpu_render_dev(ctx: *mut Ctx, cam: *const Camera, width: u32, height: u32, mp: *const MarchParams,
                         d_hit: *mut u8, d_depth: *mut f32, d_normal_xyz: *mut f32, d_prim: *mut u32,
                         stream: *mut c_void) -> c_int;
    pub fn sdfgpu_status_string(status: c_int) -> *const c_char;
    pub fn sdfgpu_sync(ctx: *mut Ctx) -> c_int;
    /// the host half of an upload alone (validation + compaction, no GPU touched)
    pub fn sdfgpu_compact_host(dt: *const SdfOperationBlock, ut: *const SdfOperationUptreeBlock, buffer_len: u32,
                               layout: u32, bounds: u32, info: *mut TreeInfo, errbuf: *mut c_char, errlen: usize) -> c_int;
    pub fn sdfgpu_compact_host_ext(dt: *const SdfOperationBlock, ut: *const SdfOperationUptreeBlock, buffer_len: u32,
                                   layout: u32, bounds: u32, extensions: u32, info: *mut TreeInfo,
                                   errbuf: *mut c_char, errlen: usize) -> c_int;
    /// tile-compact planes of one share -> row-major frame (the single-share form of sdfgpu_untile_gathered_dev)
    pub fn sdfgpu_untile_dev(ctx: *mut Ctx, width: u32, height: u32, tile_first: u32, tile_stride: u32, tile_count: u32,
                             d_hit_t: *const u8, d_depth_t: *const f32, d_normal_t: *const f32, d_prim_t: *const u32,
                             d_hit: *mut u8, d_depth: *mut f32, d_normal_xyz: *mut f32, d_prim: *mut u32,
                             stream: *mut c_void) -> c_int;
    // work counters of the instrumented kernels (what bench.py's roofline units come from)
    pub fn sdfgpu_render_stats(ctx: *mut Ctx, cam: *const Camera, width: u32, height: u32, mp: *const MarchParams,
                               out: *mut Stats) -> c_int;
    pub fn sdfgpu_nearest_stats_dev(ctx: *mut Ctx, d_pts_xyzw: *const f32, n: usize, out: *mut Stats) -> c_int;
    pub fn sdfgpu_knn_stats_dev(ctx: *mut Ctx, d_pts_xyzw: *const f32, n: usize, k: u32, out: *mut Stats) -> c_int;
    fn sdfgpu_set_option(ctx: *mut Ctx, option: c_int, value: f64) -> c_int;
    fn sdfgpu_camera_look_at(eye: *const f32, target: *const f32, up: *const f32, fov_y: f32, width: u32,
                             height: u32, out: *mut Camera);
}

#[derive(Debug)]
pub struct Error(pub i32, pub String);

/// A scene tree resident on one GPU.  `!Sync`: a context is not thread-safe.
pub struct GpuTree {
    ctx: *mut Ctx,
}

pub struct Frame {
    pub width: u32,
    pub height: u32,
    pub hit: Vec<u8>,
    pub depth: Vec<f32>,
    pub normal: Vec<[f32; 3]>,
    pub prim: Vec<u32>,
}

impl GpuTree {
    fn fail(&self, rc: c_int) -> Error {
        let msg = unsafe { CStr::from_ptr(sdfgpu_last_error(self.ctx)) }.to_string_lossy().into_owned();
        Error(rc, msg)
    }

    /// `finalize().expanded().make_buffer()` of the crate, then this.
    pub fn upload(device: i32, tree: &SdfTreeBuffer) -> Result<Self, Error> {
        let mut ctx = std::ptr::null_mut();
        let rc = unsafe { sdfgpu_create(device, &mut ctx) };
        if rc != 0 {
            return Err(Error(rc, "sdfgpu_create failed".into()));
        }
        let this = GpuTree { ctx };
        let rc = unsafe {
            sdfgpu_upload_tree(ctx, tree.downtree_buffer.as_ptr(), tree.uptree_buffer.as_ptr(), tree.buffer_len,
                               LAYOUT_GLAM_SSE2, BOUNDS_AUTO)
        };
        if rc != 0 {
            return Err(this.fail(rc)); // `this` drops and frees the context
        }
        Ok(this)
    }

    /// The same with opt-in extensions (`EXT_*` bits); 0 = `upload`.
    pub fn upload_ext(device: i32, tree: &SdfTreeBuffer, extensions: u32) -> Result<Self, Error> {
        let mut ctx = std::ptr::null_mut();
        let rc = unsafe { sdfgpu_create(device, &mut ctx) };
        if rc != 0 {
            return Err(Error(rc, "sdfgpu_create failed".into()));
        }
        let this = GpuTree { ctx };
        let rc = unsafe {
            sdfgpu_upload_tree_ext(ctx, tree.downtree_buffer.as_ptr(), tree.uptree_buffer.as_ptr(), tree.buffer_len,
                                   LAYOUT_GLAM_SSE2, BOUNDS_AUTO, extensions)
        };
        if rc != 0 {
            return Err(this.fail(rc));
        }
        Ok(this)
    }

    pub fn info(&self) -> TreeInfo {
        let mut i = TreeInfo::default();
        unsafe { sdfgpu_tree_info_get(self.ctx, &mut i) };
        i
    }

    /// Batch form of `faux_shader::nearest_neighbor(&tree, point)` (faux_shader.rs:96):
    /// the same distances, plus the pre-order ordinal of the primitive that attains each.
    pub fn nearest_neighbor(&self, points: &[Vec4]) -> Result<(Vec<f32>, Vec<u32>), Error> {
        let mut dist = vec![0f32; points.len()];
        let mut prim = vec![NO_PRIM; points.len()];
        let rc = unsafe {
            sdfgpu_nearest(self.ctx, points.as_ptr() as *const f32, points.len(), dist.as_mut_ptr(), prim.as_mut_ptr())
        };
        if rc != 0 { Err(self.fail(rc)) } else { Ok((dist, prim)) }
    }

    /// k nearest primitives per point (README.md:7), row-major `[n][k]`.
    pub fn knn(&self, points: &[Vec4], k: u32) -> Result<(Vec<f32>, Vec<u32>), Error> {
        let n = points.len() * k as usize;
        let mut dist = vec![0f32; n];
        let mut prim = vec![NO_PRIM; n];
        let rc = unsafe {
            sdfgpu_knn(self.ctx, points.as_ptr() as *const f32, points.len(), k, dist.as_mut_ptr(), prim.as_mut_ptr())
        };
        if rc != 0 { Err(self.fail(rc)) } else { Ok((dist, prim)) }
    }

    /// MINDIST / MAXDIST of one box (faux_shader.rs:39-58) at many points.
    pub fn box_bounds(&self, bbox: &SdfBoundingBoxBlock, points: &[Vec4]) -> Result<(Vec<f32>, Vec<f32>), Error> {
        let mut lo = vec![0f32; points.len()];
        let mut hi = vec![0f32; points.len()];
        let rc = unsafe {
            sdfgpu_box_bounds(self.ctx, bbox, points.as_ptr() as *const f32, points.len(), lo.as_mut_ptr(), hi.as_mut_ptr())
        };
        if rc != 0 { Err(self.fail(rc)) } else { Ok((lo, hi)) }
    }

    /// Sphere-traced frame with normals (march loop + src/main__.rs:151-161).
    pub fn render(&self, eye: Vec3, target: Vec3, up: Vec3, fov_y: f32, width: u32, height: u32,
                  mp: &MarchParams) -> Result<Frame, Error> {
        let mut cam = Camera::default();
        unsafe {
            sdfgpu_camera_look_at(eye.as_ref().as_ptr(), target.as_ref().as_ptr(), up.as_ref().as_ptr(), fov_y,
                                  width, height, &mut cam)
        };
        let n = (width as usize) * (height as usize);
        let mut f = Frame { width, height, hit: vec![0; n], depth: vec![0.0; n], normal: vec![[0.0; 3]; n], prim: vec![NO_PRIM; n] };
        let rc = unsafe {
            sdfgpu_render(self.ctx, &cam, width, height, mp, f.hit.as_mut_ptr(), f.depth.as_mut_ptr(),
                          f.normal.as_mut_ptr() as *mut f32, f.prim.as_mut_ptr())
        };
        if rc != 0 { Err(self.fail(rc)) } else { Ok(f) }
    }

    pub fn set_option(&self, option: i32, value: f64) -> Result<(), Error> {
        let rc = unsafe { sdfgpu_set_option(self.ctx, option, value) };
        if rc != 0 { Err(self.fail(rc)) } else { Ok(()) }
    }

    /// For callers that own device memory and a HIP stream (asynchronous).
    ///
    /// # Safety
    /// The pointers must be device pointers of at least width*height elements (normal: 3x).
    pub unsafe fn render_dev(&self, cam: &Camera, width: u32, height: u32, mp: &MarchParams, d_hit: *mut u8,
                             d_depth: *mut f32, d_normal: *mut f32, d_prim: *mut u32, stream: *mut c_void)
                             -> Result<(), Error> {
        let rc = sdfgpu_render_dev(self.ctx, cam, width, height, mp, d_hit, d_depth, d_normal, d_prim, stream);
        if rc != 0 { Err(self.fail(rc)) } else { Ok(()) }
    }
}

impl Drop for GpuTree {
    fn drop(&mut self) {
        unsafe { sdfgpu_destroy(self.ctx) }
    }
}

#[cfg(test)]
mod tests {
    //! The crate's own test idea (node.rs:506-580) run against the GPU: one transformed
    //! sphere, the query must be |p - c| - r.
    use super::*;
    use bevy::prelude::Transform;
    use sdf::elements::SdfSphere;
    use sdf::node::SdfBuilder;

    #[test]
    fn single_sphere_matches_the_cpu_path() {
        let node = SdfBuilder::primitive(SdfSphere { radius: 1.5 })
            .transform(Transform::from_xyz(3.0, -2.0, 5.0))
            .finalize();
        let buffer = node.expanded().make_buffer();
        let gpu = GpuTree::upload(0, &buffer).expect("upload");
        let pts: Vec<Vec4> = (0..64).map(|i| Vec4::new(i as f32 * 0.3 - 9.0, 1.0, -4.0 + i as f32 * 0.1, 1.0)).collect();
        let (dist, prim) = gpu.nearest_neighbor(&pts).expect("query");
        for (p, (d, i)) in pts.iter().zip(dist.iter().zip(prim.iter())) {
            let cpu = sdf::faux_shader::nearest_neighbor(&buffer, *p);
            assert_eq!(d.to_bits(), cpu.to_bits());
            assert_eq!(*i, 0);
        }
    }
}
