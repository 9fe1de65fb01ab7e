//! `HipBruteForceSimulation` / `HipBarnesHutSimulation`: the reference's `Simulation` trait
//! (src/shared.rs:80-97) implemented over include/nbody_hip.h.
//!
//! UNTESTED SOURCE: there is no Rust toolchain in the build image.  The C-ABI side of every call
//! below is covered by tests/ through the ctypes binding and the C++ mirror
//! (nbody-llm_amd/host/simulation.hpp), which use the same entry points in the same order.
#![feature(generic_const_exprs)]
#![allow(incomplete_features)]

use std::cell::{Cell, UnsafeCell};
use std::ffi::{c_char, c_int, c_void, CStr};

use nbody::shared::{Bounds, Integrator, LeapFrogIntegrator, PointParticle, Simulation, SimulationSettings, AABB};

type P = PointParticle<f32, 3>;
type I = LeapFrogIntegrator<f32, 3, P>;

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
    reserved: i32,
}

#[repr(C)]
struct NbodyHandle {
    _private: [u8; 0],
}

extern "C" {
    fn nbody_create(cfg: *const NbodyConfig, out: *mut *mut NbodyHandle) -> c_int;
    fn nbody_destroy(h: *mut NbodyHandle);
    fn nbody_clone(h: *const NbodyHandle, out: *mut *mut NbodyHandle) -> c_int;
    fn nbody_upload(h: *mut NbodyHandle, aos: *const c_void, n: usize, stride: usize) -> c_int;
    fn nbody_download(h: *mut NbodyHandle, aos: *mut c_void, cap: usize, stride: usize, n_out: *mut usize) -> c_int;
    fn nbody_count(h: *mut NbodyHandle, n_out: *mut usize) -> c_int;
    fn nbody_add_point(h: *mut NbodyHandle, particle: *const c_void) -> c_int;
    fn nbody_remove_point(h: *mut NbodyHandle, index: usize) -> c_int;
    fn nbody_set_settings(h: *mut NbodyHandle, g: f32, g_soft: f32, dt: f32, theta2: f32) -> c_int;
    fn nbody_set_bounds(h: *mut NbodyHandle, center: *const f32, width: f32) -> c_int;
    fn nbody_init(h: *mut NbodyHandle) -> c_int;
    fn nbody_step_by(h: *mut NbodyHandle, dt: f32) -> c_int;
    fn nbody_update_forces(h: *mut NbodyHandle) -> c_int;
    fn nbody_last_error(h: *const NbodyHandle) -> *const c_char;
}

const NBODY_BRUTE_FORCE: i32 = 0;
const NBODY_BARNES_HUT: i32 = 1;
const NBODY_MATH_FAST: i32 = 1;

/// One GPU-resident simulation.  `METHOD` selects which reference solver it stands in for.
pub struct HipSimulation<const METHOD: i32> {
    handle: *mut NbodyHandle,
    /// host mirror handed out by `get_points(&self)`; refreshed lazily (the trait borrows `&self`,
    /// hence the interior mutability)
    points: UnsafeCell<Vec<P>>,
    dirty: Cell<bool>,
    integrator: I,
    bounds: Bounds<f32, 3>,
    settings: SimulationSettings<f32>,
    settings_dirty: bool,
    elapsed: f32,
}

pub type HipBruteForceSimulation = HipSimulation<NBODY_BRUTE_FORCE>;
pub type HipBarnesHutSimulation = HipSimulation<NBODY_BARNES_HUT>;

impl<const METHOD: i32> HipSimulation<METHOD> {
    fn check(&self, rc: c_int) {
        if rc != 0 {
            // the reference's trait is infallible (it unwraps); keep that contract
            let msg = unsafe { CStr::from_ptr(nbody_last_error(self.handle)) }.to_string_lossy().into_owned();
            panic!("nbody_hip error {rc}: {msg}");
        }
    }

    fn push_settings(&mut self) {
        if self.settings_dirty {
            let s = &self.settings;
            self.check(unsafe { nbody_set_settings(self.handle, s.g, s.g_soft, s.dt, s.theta2) });
            self.settings_dirty = false;
        }
    }

    fn create(points: &[P], bounds: &Bounds<f32, 3>) -> *mut NbodyHandle {
        let cfg = NbodyConfig {
            struct_size: std::mem::size_of::<NbodyConfig>() as u32,
            method: METHOD,
            math_mode: NBODY_MATH_FAST,
            leaf_mode: 0, // NBODY_LEAF_REFERENCE (src/manual); 1 = NBODY_LEAF_DIRECT (the src/llm walk)
            device: -1,
            rank: 0,
            world_size: 1,
            host_threads: 0,
            capacity: (points.len().max(1) * 2) as u64, // headroom for add_point
            tree_build: 0,
            reserved: 0,
        };
        let mut h: *mut NbodyHandle = std::ptr::null_mut();
        let rc = unsafe { nbody_create(&cfg, &mut h) };
        if rc != 0 {
            let msg = unsafe { CStr::from_ptr(nbody_last_error(std::ptr::null())) }.to_string_lossy().into_owned();
            panic!("nbody_create failed ({rc}): {msg}");
        }
        let c = bounds.center();
        let center = [c[0], c[1], c[2]];
        unsafe {
            assert_eq!(nbody_set_bounds(h, center.as_ptr(), bounds.width), 0);
            // PointParticle<f32,3> is #[repr(C)] {position, velocity, acceleration, mass}: 40 bytes
            assert_eq!(nbody_upload(h, points.as_ptr() as *const c_void, points.len(), std::mem::size_of::<P>()), 0);
        }
        h
    }
}

impl<const METHOD: i32> Drop for HipSimulation<METHOD> {
    fn drop(&mut self) {
        unsafe { nbody_destroy(self.handle) }
    }
}

impl<const METHOD: i32> Clone for HipSimulation<METHOD> {
    fn clone(&self) -> Self {
        let mut h: *mut NbodyHandle = std::ptr::null_mut();
        self.check(unsafe { nbody_clone(self.handle, &mut h) });
        Self {
            handle: h,
            points: UnsafeCell::new(Vec::new()),
            dirty: Cell::new(true),
            integrator: self.integrator.clone(),
            bounds: self.bounds,
            settings: self.settings.clone(),
            settings_dirty: true,
            elapsed: self.elapsed,
        }
    }
}

impl<const METHOD: i32> Simulation<f32, 3, P, I> for HipSimulation<METHOD> {
    fn new(points: Vec<P>, integrator: I, bounds: Bounds<f32, 3>) -> Self {
        let handle = Self::create(&points, &bounds);
        Self {
            handle,
            points: UnsafeCell::new(points),
            dirty: Cell::new(false),
            integrator,
            bounds,
            settings: SimulationSettings::default(),
            settings_dirty: true,
            elapsed: 0.0,
        }
    }

    fn init(&mut self) {
        self.integrator.init();
        self.elapsed = 0.0;
        self.check(unsafe { nbody_init(self.handle) });
    }

    fn step_by(&mut self, dt: f32) {
        self.push_settings();
        self.check(unsafe { nbody_step_by(self.handle, dt) });
        self.elapsed += dt;
        self.dirty.set(true);
    }

    fn update_forces(&mut self) {
        self.push_settings();
        self.check(unsafe { nbody_update_forces(self.handle) });
        self.dirty.set(true);
    }

    fn add_point(&mut self, point: P) {
        self.check(unsafe { nbody_add_point(self.handle, &point as *const P as *const c_void) });
        self.dirty.set(true);
    }

    fn remove_point(&mut self, index: usize) {
        self.check(unsafe { nbody_remove_point(self.handle, index) });
        self.dirty.set(true);
    }

    fn get_points(&self) -> &Vec<P> {
        // SAFETY: single-threaded use (the reference calls its simulation from one thread); the
        // vector is only replaced here, never while a previously returned borrow can be live
        // across a &mut self call.
        let v = unsafe { &mut *self.points.get() };
        if self.dirty.get() {
            let mut n: usize = 0;
            self.check(unsafe { nbody_count(self.handle, &mut n) });
            v.resize(n, P::new([0.0; 3].into(), [0.0; 3].into(), 0.0, 0.0));
            self.check(unsafe {
                nbody_download(self.handle, v.as_mut_ptr() as *mut c_void, n, std::mem::size_of::<P>(), &mut n)
            });
            v.truncate(n);
            self.dirty.set(false);
        }
        v
    }

    fn elapsed(&self) -> f32 {
        self.elapsed
    }

    fn settings(&self) -> &SimulationSettings<f32> {
        &self.settings
    }

    fn settings_mut(&mut self) -> &mut SimulationSettings<f32> {
        self.settings_dirty = true;
        &mut self.settings
    }
}
