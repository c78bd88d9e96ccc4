//! Drop-in for `jgrodzki/radix_sort`'s sort path: the two public traits of the reference
//! (`src/radix_sort/mod.rs:18-20`, `src/radix_sort/radix_digits.rs:1-5`) with the body of
//! `radix_sort()` replaced by one call into the MI355X library (`include/rsx.h`).
//!
//! `get_digit` is arbitrary user code, so the device cannot run it; device dispatch keys off
//! an additional associated const, `RSX_KEY`, which the 14 built-in key kinds set and the tuple
//! impl forwards.  A type that leaves it `None`, or whose size has no device kernel, takes the CPU
//! path `cpu_radix_sort` (the reference's algorithm, restated below), so everything the reference
//! accepts still compiles AND sorts.
//!
//! STATUS: this crate has never been compiled -- the build image has no cargo/rustc -- so it is
//! text, not a tested artefact; every C call it makes is exercised by the C++ mirror
//! (`radix_sort_amd/cxx/radix_sort.hpp`) and the ctypes harness.  Parity with the reference's own
//! binary is unpinned for the same reason.
#![allow(clippy::missing_safety_doc)]
use core::ffi::{c_char, c_int, c_void};

#[repr(C)]
#[derive(Clone, Copy, Debug, PartialEq, Eq)]
pub struct RsxLayout {
    pub elem_bytes: u32,
    pub key_offset: u32,
    pub key_bytes: u32,
    pub key_kind: u32,
}
pub const RSX_KEY_UNSIGNED: u32 = 0;
pub const RSX_KEY_SIGNED: u32 = 1;
pub const RSX_KEY_FLOAT: u32 = 2;

#[repr(C)]
pub struct RsxCtx {
    _private: [u8; 0],
}

extern "C" {
    pub fn rsx_ctx_create(device: c_int, out: *mut *mut RsxCtx) -> c_int;
    pub fn rsx_ctx_destroy(ctx: *mut RsxCtx) -> c_int;
    pub fn rsx_sort_host(ctx: *mut RsxCtx, data: *mut c_void, n: usize, layout: *const RsxLayout) -> c_int;
    pub fn rsx_sort_device(ctx: *mut RsxCtx, d_data: *mut c_void, d_tmp: *mut c_void, n: usize,
                           layout: *const RsxLayout, stream: *mut c_void) -> c_int;
    pub fn rsx_sort_sharded(ctxs: *const *mut RsxCtx, ndev: u32, d_slices: *const *mut c_void,
                            d_tmps: *const *mut c_void, n_per_dev: *const usize,
                            layout: *const RsxLayout) -> c_int;
    pub fn rsx_strerror(status: c_int) -> *const c_char;
}

/// Key descriptor of a built-in key kind: (key_bytes, key_kind); offset and element size are
/// filled in per element type.
#[derive(Clone, Copy)]
pub struct RsxKey {
    pub key_bytes: u32,
    pub key_kind: u32,
    pub key_offset: u32,
}

/// `src/radix_sort/radix_digits.rs:1-5`, plus the device descriptor.
pub trait RadixDigits: Send + Sync {
    const NUMBER_OF_DIGITS: u8;
    /// `Some` for keys the device knows how to map; `None` => CPU path.
    const RSX_KEY: Option<RsxKey> = None;
    fn get_digit(&self, index: u8) -> u8;
}

macro_rules! unsigned_impl { ($($t:ty),*) => {$(
    impl RadixDigits for $t {
        const NUMBER_OF_DIGITS: u8 = core::mem::size_of::<$t>() as u8;
        const RSX_KEY: Option<RsxKey> = Some(RsxKey { key_bytes: core::mem::size_of::<$t>() as u32, key_kind: RSX_KEY_UNSIGNED, key_offset: 0 });
        fn get_digit(&self, index: u8) -> u8 { (*self >> (index as u32 * 8)) as u8 }
    }
)*}}
macro_rules! signed_impl { ($($t:ty),*) => {$(
    impl RadixDigits for $t {
        const NUMBER_OF_DIGITS: u8 = core::mem::size_of::<$t>() as u8;
        const RSX_KEY: Option<RsxKey> = Some(RsxKey { key_bytes: core::mem::size_of::<$t>() as u32, key_kind: RSX_KEY_SIGNED, key_offset: 0 });
        fn get_digit(&self, index: u8) -> u8 { ((*self ^ <$t>::MIN) >> (index as u32 * 8)) as u8 }
    }
)*}}
unsigned_impl!(u8, u16, u32, u64, u128, usize);
signed_impl!(i8, i16, i32, i64, i128, isize);

impl RadixDigits for f32 {
    const NUMBER_OF_DIGITS: u8 = 4;
    const RSX_KEY: Option<RsxKey> = Some(RsxKey { key_bytes: 4, key_kind: RSX_KEY_FLOAT, key_offset: 0 });
    fn get_digit(&self, index: u8) -> u8 {
        let mut b = self.to_bits() as i32;
        b ^= (b >> 31) | i32::MIN;
        (b as u32 >> (index as u32 * 8)) as u8
    }
}
impl RadixDigits for f64 {
    const NUMBER_OF_DIGITS: u8 = 8;
    const RSX_KEY: Option<RsxKey> = Some(RsxKey { key_bytes: 8, key_kind: RSX_KEY_FLOAT, key_offset: 0 });
    fn get_digit(&self, index: u8) -> u8 {
        let mut b = self.to_bits() as i64;
        b ^= (b >> 63) | i64::MIN;
        (b as u64 >> (index as u32 * 8)) as u8
    }
}
/// `(T, U)`: key is `.0`, payload opaque (radix_digits.rs:126-136).  Rust tuple layout is not
/// ABI-stable, so the key offset is measured, not assumed.
impl<T: RadixDigits + Send + Sync, U: Send + Sync> RadixDigits for (T, U) {
    const NUMBER_OF_DIGITS: u8 = T::NUMBER_OF_DIGITS;
    const RSX_KEY: Option<RsxKey> = match T::RSX_KEY {
        Some(k) => Some(RsxKey { key_bytes: k.key_bytes, key_kind: k.key_kind,
                                 key_offset: core::mem::offset_of!((T, U), 0) as u32 + k.key_offset }),
        None => None,
    };
    fn get_digit(&self, index: u8) -> u8 { self.0.get_digit(index) }
}

/// `src/radix_sort/mod.rs:18-20`.
pub trait RadixSort<T: RadixDigits> {
    fn radix_sort(&mut self);
}

fn with_ctx<R>(f: impl FnOnce(*mut RsxCtx) -> R) -> R {
    use std::sync::{Mutex, OnceLock};
    struct Ctx(*mut RsxCtx);
    unsafe impl Send for Ctx {}
    static CTX: OnceLock<Mutex<Ctx>> = OnceLock::new();
    let m = CTX.get_or_init(|| {
        let mut p = core::ptr::null_mut();
        let rc = unsafe { rsx_ctx_create(-1, &mut p) };
        assert!(rc == 0, "rsx_ctx_create failed: {rc}"); // the reference panics too (mod.rs:68)
        Mutex::new(Ctx(p))
    });
    let g = m.lock().unwrap();
    f(g.0)
}

impl<T: RadixDigits> RadixSort<T> for [T] {
    fn radix_sort(&mut self) {
        if self.len() <= 1 {
            return; // the reference panics on an empty slice (mod.rs:66-70,92): nothing to sort
        }
        match T::RSX_KEY {
            Some(k) if matches!(core::mem::size_of::<T>(), 1 | 2 | 4 | 8 | 12 | 16 | 24 | 32) => {
                let layout = RsxLayout { elem_bytes: core::mem::size_of::<T>() as u32, key_offset: k.key_offset,
                                         key_bytes: k.key_bytes, key_kind: k.key_kind };
                let rc = with_ctx(|ctx| unsafe {
                    rsx_sort_host(ctx, self.as_mut_ptr() as *mut c_void, self.len(), &layout)
                });
                assert!(rc == 0, "rsx_sort_host failed: {rc}"); // panic like mod.rs:106
            }
            _ => cpu_radix_sort(self),
        }
    }
}

/// CPU path, for the element types the device library has no kernel for: a user key type with its own
/// `get_digit` (`RSX_KEY == None`) or an element size outside {1, 2, 4, 8, 12, 16, 24, 32}.  Same
/// algorithm as the reference's body (`src/radix_sort/mod.rs:62-175`), restated: one chunk per hardware
/// thread, per-chunk digit counts, digit-major / chunk-minor running sum, scatter through 96-element
/// staging runs per digit, ping-pong between the slice and a scratch buffer, copy-back after an odd
/// number of passes.  Elements are moved bitwise (no `Copy` bound), as in `mod.rs:133-140`.
/// Same output as the device path on the types both accept: a stable sort by mapped key has one answer.
fn cpu_radix_sort<T: RadixDigits>(data: &mut [T]) {
    use core::mem::MaybeUninit;
    use core::ptr::copy_nonoverlapping;
    const STAGE: usize = 96; // mod.rs:64

    struct Raw<T>(*mut T);
    impl<T> Clone for Raw<T> {
        fn clone(&self) -> Self {
            Raw(self.0)
        }
    }
    impl<T> Copy for Raw<T> {}
    // the chunks write disjoint ranges of the destination (their cursors come from one running sum)
    unsafe impl<T> Send for Raw<T> {}
    unsafe impl<T> Sync for Raw<T> {}

    let n = data.len();
    let threads = std::thread::available_parallelism().map(|v| v.get()).unwrap_or(1);
    let per_chunk = (n + threads - 1) / threads; // n >= 2 here
    let mut scratch: Vec<MaybeUninit<T>> = Vec::with_capacity(n);
    // SAFETY: MaybeUninit needs no initialisation; every slot is written by pass 0 before it is read.
    unsafe { scratch.set_len(n) };
    let slice_ptr = data.as_mut_ptr();
    let scratch_ptr = scratch.as_mut_ptr() as *mut T;

    for digit in 0..T::NUMBER_OF_DIGITS {
        let (src_ptr, dst_ptr) = if digit % 2 == 0 { (slice_ptr, scratch_ptr) } else { (scratch_ptr, slice_ptr) };
        // SAFETY: both buffers hold n initialised elements whenever they are the source of a pass.
        let src: &[T] = unsafe { core::slice::from_raw_parts(src_ptr as *const T, n) };
        // count (mod.rs:90-109)
        let mut cursors: Vec<[usize; 256]> = std::thread::scope(|scope| {
            let workers: Vec<_> = src
                .chunks(per_chunk)
                .map(|chunk| {
                    scope.spawn(move || {
                        let mut counts = [0usize; 256];
                        for element in chunk {
                            counts[element.get_digit(digit) as usize] += 1;
                        }
                        counts
                    })
                })
                .collect();
            workers.into_iter().map(|w| w.join().expect("count worker panicked")).collect()
        });
        // prefix: digit-major, chunk-minor (mod.rs:110-120) -- this order is what makes the pass stable
        let mut running = 0usize;
        for value in 0..256 {
            for chunk_counts in cursors.iter_mut() {
                let count = chunk_counts[value];
                chunk_counts[value] = running;
                running += count;
            }
        }
        // scatter through staging runs (mod.rs:121-168)
        let dst = Raw(dst_ptr);
        std::thread::scope(|scope| {
            for (chunk, mut cursor) in src.chunks(per_chunk).zip(cursors.into_iter()) {
                scope.spawn(move || {
                    let dst = dst;
                    let mut staging: Vec<MaybeUninit<T>> = Vec::with_capacity(256 * STAGE);
                    // SAFETY: MaybeUninit; a slot is read only after it was filled.
                    unsafe { staging.set_len(256 * STAGE) };
                    let stage = staging.as_mut_ptr() as *mut T;
                    let mut filled = [0usize; 256];
                    for element in chunk {
                        let value = element.get_digit(digit) as usize;
                        // SAFETY: bitwise moves between disjoint buffers; indices are in range by the counts.
                        unsafe {
                            copy_nonoverlapping(element as *const T, stage.add(value * STAGE + filled[value]), 1);
                        }
                        filled[value] += 1;
                        if filled[value] == STAGE {
                            unsafe { copy_nonoverlapping(stage.add(value * STAGE), dst.0.add(cursor[value]), STAGE) };
                            cursor[value] += STAGE;
                            filled[value] = 0;
                        }
                    }
                    for value in 0..256 {
                        if filled[value] > 0 {
                            unsafe { copy_nonoverlapping(stage.add(value * STAGE), dst.0.add(cursor[value]), filled[value]) };
                        }
                    }
                });
            }
        });
    }
    if T::NUMBER_OF_DIGITS % 2 == 1 {
        // odd number of passes: the result sits in the scratch buffer (mod.rs:170-174)
        unsafe { copy_nonoverlapping(scratch_ptr as *const T, slice_ptr, n) };
    }
    // `scratch` holds MaybeUninit<T>: dropping it runs no destructor, the elements live on in `data`
}
