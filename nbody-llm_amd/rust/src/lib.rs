//! `HipBruteForceSimulation<F>` / `HipBarnesHutSimulation<F>`: the reference's `Simulation` trait
//! (src/shared.rs:80-97) and its `Renderable` trait (src/render/mod.rs:12-15) implemented over
//! include/nbody_hip.h, for F = f32 and F = f64 (the reference's driver runs f64, src/main.rs:52-105).
//!
//! UNCOMPILED SOURCE: there is no Rust toolchain in the build image (and the reference needs nightly plus
//! network access to build).  The C-ABI side of every call below is covered by tests/ through the ctypes
//! binding and the C++ mirror (nbody-llm_amd/host/simulation.hpp), which use the same entry points in the
//! same order; what has not been checked by a compiler is this file.
//!
//! Drop-in use (src/main.rs:97-101):
//! ```ignore
//! let mut sim = nbody_hip::HipBarnesHutSimulation::<f64>::new(points, LeapFrogIntegrator::new(),
//!                                                             Bounds::new([0.0, 0.0, 0.0].into(), box_width));
//! ```
#![feature(generic_const_exprs)]
#![allow(incomplete_features)]

use std::cell::{Cell, UnsafeCell};
use std::ffi::{c_char, c_int, c_void, CStr};

// the reference's library target is `nlib` (Cargo.toml: [lib] name = "nlib")
use nlib::shared::{AABB, Bounds, Float, Integrator, LeapFrogIntegrator, Particle, PointParticle, Simulation, SimulationSettings};

#[cfg(feature = "render")]
use nlib::render::{BufferWrapper, Context, PipelineType, Renderable, Renderer};

#[repr(C)]
struct NbodyConfig {
    struct_size: u32,
    method: i32,
    math_mode: i32,
    leaf_mode: i32,
    device: i32,
    rank: i32,
    world_size: i32,
    host_threads: i32,
    capacity: u64,
    tree_build: i32,
    dtype: i32,
    shard_mode: i32, // NBODY_SHARD_INDEX (0); NBODY_SHARD_SPATIAL (1) is the multi-GPU halo-exchange mode
    reserved: i32,
}

#[repr(C)]
pub struct NbodyHandle {
    _private: [u8; 0],
}

// edition 2024: extern blocks are `unsafe extern`; every item is unsafe to call
unsafe extern "C" {
    fn nbody_create(cfg: *const NbodyConfig, out: *mut *mut NbodyHandle) -> c_int;
    fn nbody_destroy(h: *mut NbodyHandle);
    fn nbody_clone(h: *const NbodyHandle, out: *mut *mut NbodyHandle) -> c_int;
    fn nbody_upload(h: *mut NbodyHandle, aos: *const c_void, n: usize, stride: usize) -> c_int;
    fn nbody_download(h: *mut NbodyHandle, aos: *mut c_void, cap: usize, stride: usize, n_out: *mut usize) -> c_int;
    fn nbody_count(h: *mut NbodyHandle, n_out: *mut usize) -> c_int;
    fn nbody_add_point(h: *mut NbodyHandle, particle: *const c_void) -> c_int;
    fn nbody_remove_point(h: *mut NbodyHandle, index: usize) -> c_int;
    fn nbody_set_settings(h: *mut NbodyHandle, g: f32, g_soft: f32, dt: f32, theta2: f32) -> c_int;
    fn nbody_set_settings_f64(h: *mut NbodyHandle, g: f64, g_soft: f64, dt: f64, theta2: f64) -> c_int;
    fn nbody_set_bounds(h: *mut NbodyHandle, center: *const f32, width: f32) -> c_int;
    fn nbody_set_bounds_f64(h: *mut NbodyHandle, center: *const f64, width: f64) -> c_int;
    fn nbody_init(h: *mut NbodyHandle) -> c_int;
    fn nbody_step_by(h: *mut NbodyHandle, dt: f32) -> c_int;
    fn nbody_step_by_f64(h: *mut NbodyHandle, dt: f64) -> c_int;
    fn nbody_steps(h: *mut NbodyHandle, k: c_int) -> c_int;
    fn nbody_update_forces(h: *mut NbodyHandle) -> c_int;
    fn nbody_sync(h: *mut NbodyHandle) -> c_int;
    fn nbody_tree_export_cells(h: *mut NbodyHandle, min_max6: *mut f32, depth: *mut i32, cap: usize, n_nodes: *mut usize) -> c_int;
    fn nbody_set_tuning(h: *mut NbodyHandle, name: *const c_char, value: c_int) -> c_int;
    fn nbody_last_error(h: *const NbodyHandle) -> *const c_char;
}

pub const NBODY_BRUTE_FORCE: i32 = 0; // src/manual/brute_force.rs
pub const NBODY_BARNES_HUT: i32 = 1; // src/manual/barnes_hut.rs

/// Arithmetic of the force kernels (include/nbody_hip.h).
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
pub enum MathMode {
    /// the reference's rounding sequence (sqrt, (d*d)*d, g/r^3, partners in ascending order): bit-exact
    Strict = 0,
    /// v_rsq_f32 + FMA, partner range split over waves: <= 1e-5 relative (f32 only; f64 always runs strict)
    Fast = 1,
}

/// Barnes-Hut leaf semantics.
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
pub enum LeafMode {
    /// src/manual/barnes_hut.rs:185-203: a leaf failing the opening test contributes nothing
    Reference = 0,
    /// the walk of src/llm/barnes_hut.rs:915-997 on the same tree: such a leaf is evaluated directly
    Direct = 1,
}

/// Where the octree is built.
#[derive(Debug, Clone, Copy, PartialEq, Eq)]
pub enum TreeBuild {
    Host = 0,
    Device = 1,
    /// fast math -> device, strict math -> host (the bit-exact path)
    Auto = 2,
}

/// What `Simulation::new` cannot carry (the trait's signature is fixed): pass it to `with_options`.
#[derive(Debug, Clone, Copy)]
pub struct HipOptions {
    pub math_mode: MathMode,
    pub leaf_mode: LeafMode,
    /// Barnes-Hut tree: Auto = device for f32 fast math, host otherwise (F = f64: host = bit-exact; Device = same cells,
    /// centres of mass to their last bits, about 4x the steps per second)
    pub tree_build: TreeBuild,
    /// octree-build threads (the reference's `-t`, src/main.rs:34-35); 0 = all
    pub host_threads: i32,
    /// HIP device ordinal; -1 = LOCAL_RANK or 0
    pub device: i32,
    /// max bodies (add_point may grow up to this); 0 = twice the initial count
    pub capacity: usize,
}

impl Default for HipOptions {
    fn default() -> Self {
        Self { math_mode: MathMode::Fast, leaf_mode: LeafMode::Reference, tree_build: TreeBuild::Auto, host_threads: 0, device: -1, capacity: 0 }
    }
}

mod sealed {
    pub trait Sealed {}
    impl Sealed for f32 {}
    impl Sealed for f64 {}
}

/// The two instantiations of the reference's `Float` the library accepts, with their entry points.
pub trait HipFloat: Float + sealed::Sealed {
    const DTYPE: i32;
    unsafe fn abi_set_settings(h: *mut NbodyHandle, s: &SimulationSettings<Self>) -> c_int;
    unsafe fn abi_set_bounds(h: *mut NbodyHandle, center: [Self; 3], width: Self) -> c_int;
    unsafe fn abi_step_by(h: *mut NbodyHandle, dt: Self) -> c_int;
}

impl HipFloat for f32 {
    const DTYPE: i32 = 0; // NBODY_F32
    unsafe fn abi_set_settings(h: *mut NbodyHandle, s: &SimulationSettings<f32>) -> c_int {
        unsafe { nbody_set_settings(h, s.g, s.g_soft, s.dt, s.theta2) }
    }
    unsafe fn abi_set_bounds(h: *mut NbodyHandle, center: [f32; 3], width: f32) -> c_int {
        unsafe { nbody_set_bounds(h, center.as_ptr(), width) }
    }
    unsafe fn abi_step_by(h: *mut NbodyHandle, dt: f32) -> c_int {
        unsafe { nbody_step_by(h, dt) }
    }
}

impl HipFloat for f64 {
    const DTYPE: i32 = 1; // NBODY_F64
    unsafe fn abi_set_settings(h: *mut NbodyHandle, s: &SimulationSettings<f64>) -> c_int {
        unsafe { nbody_set_settings_f64(h, s.g, s.g_soft, s.dt, s.theta2) }
    }
    unsafe fn abi_set_bounds(h: *mut NbodyHandle, center: [f64; 3], width: f64) -> c_int {
        unsafe { nbody_set_bounds_f64(h, center.as_ptr(), width) }
    }
    unsafe fn abi_step_by(h: *mut NbodyHandle, dt: f64) -> c_int {
        unsafe { nbody_step_by_f64(h, dt) }
    }
}

type P<F> = PointParticle<F, 3>;
type I<F> = LeapFrogIntegrator<F, 3, P<F>>;

/// One GPU-resident simulation.  `METHOD` selects which reference solver it stands in for.
pub struct HipSimulation<F: HipFloat, const METHOD: i32> {
    handle: *mut NbodyHandle,
    /// host mirror handed out by `get_points(&self)`; refreshed lazily (the trait borrows `&self`,
    /// hence the interior mutability)
    points: UnsafeCell<Vec<P<F>>>,
    dirty: Cell<bool>,
    integrator: I<F>,
    bounds: Bounds<F, 3>,
    settings: SimulationSettings<F>,
    settings_dirty: bool,
    elapsed: F,
    options: HipOptions,
    #[cfg(feature = "render")]
    points_buffer: Option<BufferWrapper>,
    #[cfg(feature = "render")]
    bounds_buffer: Option<BufferWrapper>,
    /// vertex data of the last frame, reused from frame to frame (positions: 3 per body; boxes: 10 per box)
    #[cfg(feature = "render")]
    vertex_scratch: Vec<f32>,
    /// cells of the last octree as nbody_tree_export_cells delivers them (6 per node) and their depths
    #[cfg(feature = "render")]
    cell_scratch: (Vec<f32>, Vec<i32>),
}

pub type HipBruteForceSimulation<F> = HipSimulation<F, NBODY_BRUTE_FORCE>;
pub type HipBarnesHutSimulation<F> = HipSimulation<F, NBODY_BARNES_HUT>;

fn last_error(h: *const NbodyHandle) -> String {
    unsafe { CStr::from_ptr(nbody_last_error(h)) }.to_string_lossy().into_owned()
}

impl<F: HipFloat, const METHOD: i32> HipSimulation<F, METHOD> {
    fn check(&self, rc: c_int) {
        if rc != 0 {
            // the reference's trait is infallible (it unwraps); keep that contract
            panic!("nbody_hip error {rc}: {}", last_error(self.handle));
        }
    }

    fn push_settings(&mut self) {
        if self.settings_dirty {
            let rc = unsafe { F::abi_set_settings(self.handle, &self.settings) };
            self.check(rc);
            self.settings_dirty = false;
        }
    }

    fn create(points: &[P<F>], bounds: &Bounds<F, 3>, o: &HipOptions) -> *mut NbodyHandle {
        let cfg = NbodyConfig {
            struct_size: std::mem::size_of::<NbodyConfig>() as u32,
            method: METHOD,
            math_mode: o.math_mode as i32,
            leaf_mode: o.leaf_mode as i32,
            device: o.device,
            rank: 0,
            world_size: 1,
            host_threads: o.host_threads,
            capacity: if o.capacity > 0 { o.capacity as u64 } else { (points.len().max(1) * 2) as u64 }, // headroom for add_point
            tree_build: o.tree_build as i32,
            dtype: F::DTYPE,
            shard_mode: 0,
            reserved: 0,
        };
        let mut h: *mut NbodyHandle = std::ptr::null_mut();
        let rc = unsafe { nbody_create(&cfg, &mut h) };
        if rc != 0 {
            panic!("nbody_create failed ({rc}): {}", last_error(std::ptr::null()));
        }
        let c = bounds.center(); // AABB::center (shared.rs:233-235)
        let rc = unsafe { F::abi_set_bounds(h, [c[0], c[1], c[2]], bounds.width) };
        assert_eq!(rc, 0, "nbody_set_bounds: {}", last_error(h));
        // PointParticle<F,3> is #[repr(C)] {position, velocity, acceleration, mass}: 40 bytes (f32) / 80 bytes (f64)
        let rc = unsafe { nbody_upload(h, points.as_ptr() as *const c_void, points.len(), std::mem::size_of::<P<F>>()) };
        assert_eq!(rc, 0, "nbody_upload: {}", last_error(h));
        h
    }

    /// `Simulation::new` with the choices the trait's signature has no room for.
    pub fn with_options(points: Vec<P<F>>, integrator: I<F>, bounds: Bounds<F, 3>, options: HipOptions) -> Self {
        let handle = Self::create(&points, &bounds, &options);
        Self {
            handle,
            points: UnsafeCell::new(points),
            dirty: Cell::new(false),
            integrator,
            bounds,
            settings: SimulationSettings::default(),
            settings_dirty: true,
            elapsed: F::from(0.0).unwrap(),
            options,
            #[cfg(feature = "render")]
            points_buffer: None,
            #[cfg(feature = "render")]
            bounds_buffer: None,
            #[cfg(feature = "render")]
            vertex_scratch: Vec::new(),
            #[cfg(feature = "render")]
            cell_scratch: (Vec::new(), Vec::new()),
        }
    }

    /// One launch-shape / scheme knob of this handle (include/nbody_hip.h nbody_set_tuning).
    pub fn set_tuning(&mut self, name: &str, value: i32) {
        let c = std::ffi::CString::new(name).expect("knob names have no interior NUL");
        let rc = unsafe { nbody_set_tuning(self.handle, c.as_ptr(), value) };
        self.check(rc);
    }

    /// k x `step()` with no host round trip in between (the headless loop of src/main.rs:119-122).
    pub fn steps(&mut self, k: usize) {
        self.push_settings();
        let rc = unsafe { nbody_steps(self.handle, k as c_int) };
        self.check(rc);
        let dt = self.settings.dt;
        for _ in 0..k {
            self.elapsed += dt;
        }
        self.dirty.set(true);
    }

    /// `step_by` with an integrator of the caller's choice -- the trait's generic parameter `I` (src/shared.rs:99-104).  The
    /// device fuses the reference's `LeapFrogIntegrator` (the only one it ships) into its kernels: that is `step_by`.  Any
    /// other `Integrator` runs here, on the host, as the unfused form of the same sequence (src/manual/brute_force.rs:84-90):
    /// pre-force on the synced vector, retain, forces on the device, after-force, `elapsed += dt`.  (The C++ mirror's
    /// `step_by_with` is this method; tests/test_cli.py shows a host leapfrog ending in the device's bits.)
    pub fn step_by_with<J: Integrator<F, 3, P<F>>>(&mut self, integrator: &mut J, dt: F) {
        self.push_settings();
        let mut pts: Vec<P<F>> = self.get_points().clone();
        integrator.integrate_pre_force(&mut pts, dt);
        let bounds = self.bounds;
        pts.retain(|p| bounds.contains(p.position())); // src/shared.rs:210-212: inclusive walls, NaN dropped, order kept
        self.upload_points(&pts);
        let rc = unsafe { nbody_update_forces(self.handle) };
        self.check(rc);
        self.dirty.set(true);
        let mut pts: Vec<P<F>> = self.get_points().clone();
        integrator.integrate_after_force(&mut pts, dt);
        self.upload_points(&pts);
        self.elapsed += dt;
    }

    fn upload_points(&mut self, pts: &[P<F>]) {
        let rc = unsafe { nbody_upload(self.handle, pts.as_ptr() as *const c_void, pts.len(), std::mem::size_of::<P<F>>()) };
        self.check(rc);
        self.dirty.set(true);
    }

    /// Blocks until everything enqueued has finished (call before reading a host clock).
    pub fn sync(&mut self) {
        let rc = unsafe { nbody_sync(self.handle) };
        self.check(rc);
    }

    pub fn options(&self) -> &HipOptions {
        &self.options
    }
}

impl<F: HipFloat, const METHOD: i32> Drop for HipSimulation<F, METHOD> {
    fn drop(&mut self) {
        unsafe { nbody_destroy(self.handle) }
    }
}

impl<F: HipFloat, const METHOD: i32> Clone for HipSimulation<F, METHOD> {
    fn clone(&self) -> Self {
        let mut h: *mut NbodyHandle = std::ptr::null_mut();
        let rc = unsafe { nbody_clone(self.handle, &mut h) };
        self.check(rc);
        Self {
            handle: h,
            points: UnsafeCell::new(Vec::new()),
            dirty: Cell::new(true),
            integrator: self.integrator.clone(),
            bounds: self.bounds,
            settings: self.settings.clone(),
            settings_dirty: true,
            elapsed: self.elapsed,
            options: self.options,
            #[cfg(feature = "render")]
            points_buffer: self.points_buffer.clone(),
            #[cfg(feature = "render")]
            bounds_buffer: self.bounds_buffer.clone(),
            #[cfg(feature = "render")]
            vertex_scratch: Vec::new(),
            #[cfg(feature = "render")]
            cell_scratch: (Vec::new(), Vec::new()),
        }
    }
}

impl<F: HipFloat, const METHOD: i32> Simulation<F, 3, P<F>, I<F>> for HipSimulation<F, METHOD> {
    fn new(points: Vec<P<F>>, integrator: I<F>, bounds: Bounds<F, 3>) -> Self {
        Self::with_options(points, integrator, bounds, HipOptions::default())
    }

    fn init(&mut self) {
        self.integrator.init();
        self.elapsed = F::from(0.0).unwrap();
        let rc = unsafe { nbody_init(self.handle) };
        self.check(rc);
    }

    fn step_by(&mut self, dt: F) {
        self.push_settings();
        let rc = unsafe { F::abi_step_by(self.handle, dt) };
        self.check(rc);
        self.elapsed += dt;
        self.dirty.set(true);
    }

    fn update_forces(&mut self) {
        self.push_settings();
        let rc = unsafe { nbody_update_forces(self.handle) };
        self.check(rc);
        self.dirty.set(true);
    }

    fn add_point(&mut self, point: P<F>) {
        let rc = unsafe { nbody_add_point(self.handle, &point as *const P<F> as *const c_void) };
        self.check(rc);
        self.dirty.set(true);
    }

    fn remove_point(&mut self, index: usize) {
        let rc = unsafe { nbody_remove_point(self.handle, index) };
        self.check(rc);
        self.dirty.set(true);
    }

    fn get_points(&self) -> &Vec<P<F>> {
        if self.dirty.get() {
            // SAFETY: the exclusive reference lives only inside this branch.  The trait hands out `&Vec` tied to `&self`,
            // and the device state only becomes newer than the mirror through a `&mut self` call (step, add_point, ...),
            // which cannot overlap a borrow returned here: when `dirty` is set no shared borrow of the vector exists.
            // Two overlapping `get_points()` calls (the reference's render loop makes them) both take the shared path
            // below; forming `&mut` on every call would alias the first borrow.
            let v: &mut Vec<P<F>> = unsafe { &mut *self.points.get() };
            let mut n: usize = 0;
            let rc = unsafe { nbody_count(self.handle, &mut n) };
            self.check(rc);
            let zero = F::from(0.0).unwrap();
            v.resize(n, P::<F>::new([zero; 3].into(), [zero; 3].into(), zero, zero));
            let rc = unsafe { nbody_download(self.handle, v.as_mut_ptr() as *mut c_void, n, std::mem::size_of::<P<F>>(), &mut n) };
            self.check(rc);
            v.truncate(n);
            self.dirty.set(false);
        }
        // SAFETY: shared access; nothing mutates the vector while `dirty` is clear
        unsafe { &*self.points.get() }
    }

    fn elapsed(&self) -> F {
        self.elapsed
    }

    fn settings(&self) -> &SimulationSettings<F> {
        &self.settings
    }

    fn settings_mut(&mut self) -> &mut SimulationSettings<F> {
        self.settings_dirty = true; // pushed to the device before the next step (src/vis.rs:148-187 edits these live)
        &mut self.settings
    }
}

/// The visualiser's side (src/vis.rs:25-30 wants `Simulation + Renderable`).  Same draw calls as the reference's two impls
/// (src/manual/brute_force.rs:105-171: bodies + the root box; src/manual/barnes_hut.rs:286-374: bodies + EVERY cell of the
/// octree, coloured by depth), fed from the device: positions through the lazily synced mirror of `get_points()`, cells
/// through `nbody_tree_export_cells`.  Vertex data is written in one pass into vectors that live across frames.
#[cfg(feature = "render")]
impl<F: HipFloat, const METHOD: i32> HipSimulation<F, METHOD> {
    const FLOATS_PER_BOX: usize = 10; // min xyz, max xyz, rgba: the AABB pipeline's instance layout

    /// 3 f32 per body into `vertex_scratch`; returns the number of bodies.
    fn stage_positions(&mut self) -> u32 {
        let mut out = std::mem::take(&mut self.vertex_scratch);
        out.clear();
        let pts = self.get_points();
        out.reserve(3 * pts.len());
        for p in pts {
            for x in p.position().iter() {
                out.push(num_traits::cast::<F, f32>(*x).unwrap());
            }
        }
        let n = pts.len() as u32;
        self.vertex_scratch = out;
        n
    }

    fn push_box(out: &mut Vec<f32>, lo: &[f32], hi: &[f32], rgba: [f32; 4]) {
        out.extend_from_slice(lo);
        out.extend_from_slice(hi);
        out.extend_from_slice(&rgba);
    }

    /// Boxes to draw into `vertex_scratch`; returns how many.  Brute force: the simulation bounds, green.  Barnes-Hut: the
    /// cells of the last tree, shaded from coarse to fine with the reference's ramp (barnes_hut.rs:327-336); before the
    /// first force pass there is no tree and the root box is drawn yellow, as the reference does for `root == None`.
    fn stage_boxes(&mut self) -> u32 {
        let mut out = std::mem::take(&mut self.vertex_scratch);
        out.clear();
        let root_lo: Vec<f32> = self.bounds.min().iter().map(|x| num_traits::cast::<F, f32>(*x).unwrap()).collect();
        let root_hi: Vec<f32> = self.bounds.max().iter().map(|x| num_traits::cast::<F, f32>(*x).unwrap()).collect();
        let mut boxes = 0u32;
        if METHOD == NBODY_BARNES_HUT {
            let mut n: usize = 0;
            let rc = unsafe { nbody_tree_export_cells(self.handle, std::ptr::null_mut(), std::ptr::null_mut(), 0, &mut n) };
            self.check(rc);
            if n > 0 {
                let (cells, depths) = &mut self.cell_scratch;
                cells.resize(6 * n, 0.0);
                depths.resize(n, 0);
                let rc = unsafe { nbody_tree_export_cells(self.handle, cells.as_mut_ptr(), depths.as_mut_ptr(), n, &mut n) };
                assert_eq!(rc, 0, "nbody_tree_export_cells: {}", last_error(self.handle));
                let deepest = depths[..n].iter().copied().max().unwrap_or(0).max(1) as f32;
                out.reserve(Self::FLOATS_PER_BOX * n);
                for (cell, depth) in cells.chunks_exact(6).zip(depths.iter()).take(n) {
                    let s = (*depth as f32) / deepest * 0.7 + 0.3;
                    Self::push_box(&mut out, &cell[..3], &cell[3..], [(1.0 - s * s) * 0.5, s * s, (1.0 - s) * 0.5, s]);
                }
                boxes = n as u32;
            } else {
                Self::push_box(&mut out, &root_lo, &root_hi, [1.0, 1.0, 0.0, 1.0]);
                boxes = 1;
            }
        } else {
            Self::push_box(&mut out, &root_lo, &root_hi, [0.0, 1.0, 0.0, 1.0]);
            boxes = 1;
        }
        self.vertex_scratch = out;
        boxes
    }
}

#[cfg(feature = "render")]
impl<F: HipFloat, const METHOD: i32> Renderable for HipSimulation<F, METHOD> {
    fn render(&mut self, renderer: &mut Renderer) {
        if let Some(mut vb) = self.points_buffer.take() {
            let bodies = self.stage_positions();
            vb.update(&renderer.context, self.vertex_scratch.as_slice());
            renderer.set_pipeline(PipelineType::Points);
            let pass = renderer.get_render_pass();
            pass.set_vertex_buffer(0, vb.buffer.slice(..));
            pass.draw(0..4, 0..bodies); // one 4-vertex strip per body
            self.points_buffer = Some(vb);
        }
        if let Some(mut vb) = self.bounds_buffer.take() {
            let boxes = self.stage_boxes();
            vb.update(&renderer.context, self.vertex_scratch.as_slice());
            renderer.set_pipeline(PipelineType::AABB);
            let pass = renderer.get_render_pass();
            pass.set_vertex_buffer(0, vb.buffer.slice(..));
            pass.draw(0..16, 0..boxes); // 16 line vertices per box
            self.bounds_buffer = Some(vb);
        }
    }

    fn render_init(&mut self, context: &Context) {
        let usage = wgpu::BufferUsages::VERTEX | wgpu::BufferUsages::COPY_DST;
        let empty: &[f32] = &[];
        self.points_buffer = Some(BufferWrapper::new(&context.device, Some("Point Buffer"), empty, usage));
        self.bounds_buffer = Some(BufferWrapper::new(&context.device, Some("Bounds Buffer"), empty, usage));
    }
}
