// radix_sort.hpp -- C++ host mirror of the reference's operator interface over the
// C-ABI of include/rsx.h (the reference is compiled Rust; no Rust toolchain exists in
// the build image, so the host side above the C-ABI is C++ here and the Rust shim a
// maintainer would add is given as text in INTEGRATION.md / rust/).
//
// Reference interface (src/radix_sort/radix_digits.rs:1-5, src/radix_sort/mod.rs:18-20):
//     pub trait RadixDigits { const NUMBER_OF_DIGITS: u8; fn get_digit(&self, index: u8) -> u8; }
//     pub trait RadixSort<T: RadixDigits> { fn radix_sort(&mut self); }      impl for [T]
// Mirror:
//     rsx::RadixDigits<T>::NUMBER_OF_DIGITS / ::get_digit(const T&, index) / ::layout()
//     rsx::radix_sort(T* data, size_t n)            // host slice, in place, blocking (mod.rs:62)
//     rsx::radix_sort(std::vector<T>& v)
//     rsx::radix_sort_device(T* d_data, T* d_tmp, size_t n, hipStream_t)   // device-resident
// Errors: the reference panics (mod.rs:68,106); here std::runtime_error is thrown.
// Empty and one-element slices return immediately (the reference panics on an empty
// slice -- chunks(0), mod.rs:66-70,92 -- there is no output to differ from).
#pragma once
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <vector>

#include "../../include/rsx.h"

namespace rsx {

// ---- key model (radix_digits.rs) ----------------------------------------------------
template <typename T, typename Enable = void>
struct RadixDigits;  // specialise for user types: provide layout() (and get_digit for host use)

namespace detail {
template <typename T>
constexpr uint32_t kind_of() {
    return std::is_floating_point<T>::value ? RSX_KEY_FLOAT : std::is_signed<T>::value ? RSX_KEY_SIGNED : RSX_KEY_UNSIGNED;
}
inline uint8_t digit_of(const unsigned char* key, uint32_t key_bytes, uint32_t kind, uint8_t index) {
    const uint32_t top = key_bytes - 1;
    uint8_t b = key[index];  // little-endian: byte `index` == (x >> 8*index) as u8
    if (kind == RSX_KEY_SIGNED) {
        if (index == top) b ^= 0x80;  // (x ^ MIN) >> ...   radix_digits.rs:55-101
    } else if (kind == RSX_KEY_FLOAT) {
        if (key[top] & 0x80) b ^= 0xFF;  // b ^= (b >> 31) | MIN   radix_digits.rs:103-124
        else if (index == top) b ^= 0x80;
    }
    return b;
}
}  // namespace detail

// u8..u64, i8..i64, f32, f64 (radix_digits.rs:7-38,47-85,95-124; usize/isize are the 64-bit ones)
template <typename T>
struct RadixDigits<T, typename std::enable_if<std::is_arithmetic<T>::value && !std::is_same<T, bool>::value &&
                                              sizeof(T) <= 8>::type> {
    static constexpr uint8_t NUMBER_OF_DIGITS = sizeof(T);
    static rsx_layout layout() { return rsx_layout{sizeof(T), 0, sizeof(T), detail::kind_of<T>()}; }
    static uint8_t get_digit(const T& x, uint8_t index) {
        unsigned char raw[sizeof(T)];
        std::memcpy(raw, &x, sizeof(T));
        return detail::digit_of(raw, sizeof(T), detail::kind_of<T>(), index);
    }
};

#if defined(__SIZEOF_INT128__)
// u128 / i128 (radix_digits.rs:39-45,87-93)
template <>
struct RadixDigits<unsigned __int128> {
    static constexpr uint8_t NUMBER_OF_DIGITS = 16;
    static rsx_layout layout() { return rsx_layout{16, 0, 16, RSX_KEY_UNSIGNED}; }
    static uint8_t get_digit(const unsigned __int128& x, uint8_t index) { return (uint8_t)(x >> (8 * index)); }
};
template <>
struct RadixDigits<__int128> {
    static constexpr uint8_t NUMBER_OF_DIGITS = 16;
    static rsx_layout layout() { return rsx_layout{16, 0, 16, RSX_KEY_SIGNED}; }
    static uint8_t get_digit(const __int128& x, uint8_t index) {
        unsigned char raw[16];
        std::memcpy(raw, &x, 16);
        return detail::digit_of(raw, 16, RSX_KEY_SIGNED, index);
    }
};
#endif

// (K, U): key `.first`, opaque payload carried along (radix_digits.rs:126-136).  The key offset
// is taken from the actual object layout, as the Rust shim must do with offset_of!.
template <typename K, typename U>
struct RadixDigits<std::pair<K, U>> {
    static constexpr uint8_t NUMBER_OF_DIGITS = RadixDigits<K>::NUMBER_OF_DIGITS;
    static rsx_layout layout() {
        // elements are moved bitwise (mod.rs:133-140 uses copy_nonoverlapping); std::pair itself is never
        // "trivially copyable" (user-provided assignment), so the requirement is put on its members
        static_assert(std::is_trivially_copyable<K>::value && std::is_trivially_copyable<U>::value,
                      "radix_sort moves elements bitwise: key and payload must be trivially copyable");
        const std::pair<K, U>* p = nullptr;
        const uint32_t off = (uint32_t)(reinterpret_cast<const char*>(&p->first) - reinterpret_cast<const char*>(p));
        rsx_layout k = RadixDigits<K>::layout();
        return rsx_layout{(uint32_t)sizeof(std::pair<K, U>), off + k.key_offset, k.key_bytes, k.key_kind};
    }
    static uint8_t get_digit(const std::pair<K, U>& x, uint8_t index) { return RadixDigits<K>::get_digit(x.first, index); }
};

// ---- the sort (mod.rs:18-20,61-176) -------------------------------------------------
class Context {
public:
    explicit Context(int device = -1) {
        const int rc = rsx_ctx_create(device, &ctx_);
        if (rc != RSX_OK) throw std::runtime_error(std::string("rsx_ctx_create: ") + rsx_strerror(rc));
    }
    ~Context() {
        if (ctx_) rsx_ctx_destroy(ctx_);
    }
    Context(const Context&) = delete;
    Context& operator=(const Context&) = delete;
    rsx_ctx* get() const { return ctx_; }
    void check(int rc, const char* what) const {
        if (rc != RSX_OK)
            throw std::runtime_error(std::string(what) + ": " + rsx_strerror(rc) + " (" + rsx_last_error(ctx_) + ")");
    }
    // rsx_ctx_check: synchronises `stream` and throws if a kernel of this context gave up a device-side
    // wait.  The stream-ordered radix_sort_device cannot report that by itself (the reference's analogue
    // is the worker panic of mod.rs:106): call this where you synchronise.
    void synchronize_and_check(void* stream = nullptr) const { check(rsx_ctx_check(ctx_, stream), "rsx_ctx_check"); }
    // rsx_ctx_set_option: force one of the bit-exact alternative kernel paths (RSX_OPT_*)
    void set_option(int option, uint64_t value) const { check(rsx_ctx_set_option(ctx_, option, value), "rsx_ctx_set_option"); }

private:
    rsx_ctx* ctx_ = nullptr;
};

inline Context& default_context() {
    static Context c(-1);
    return c;
}

// `<[T]>::radix_sort(&mut self)`: in place on a host slice, blocking.
template <typename T>
void radix_sort(T* data, size_t n, Context& ctx = default_context()) {
    const rsx_layout L = RadixDigits<T>::layout();
    ctx.check(rsx_sort_host(ctx.get(), data, n, &L), "rsx_sort_host");
}
template <typename T>
void radix_sort(std::vector<T>& v, Context& ctx = default_context()) {
    radix_sort(v.data(), v.size(), ctx);
}
// Device-resident form: `d_tmp` is the reference's `temp` (mod.rs:71-83); stream-ordered, not synchronised.
// A device-side failure surfaces at ctx.synchronize_and_check(stream) (or at the next sort on the context).
template <typename T>
void radix_sort_device(T* d_data, T* d_tmp, size_t n, void* stream = nullptr, Context& ctx = default_context()) {
    const rsx_layout L = RadixDigits<T>::layout();
    ctx.check(rsx_sort_device(ctx.get(), d_data, d_tmp, n, &L, stream), "rsx_sort_device");
}

// Multi-GPU, one process: slice g lives on the device of ctxs[g]; the concatenation of the slices
// is sorted as one array (slice = the reference's "chunk", mod.rs:66-70).  Blocking.
template <typename T>
void radix_sort_sharded(const std::vector<Context*>& ctxs, const std::vector<T*>& d_slices,
                        const std::vector<T*>& d_tmps, const std::vector<size_t>& n_per_dev) {
    if (ctxs.empty() || ctxs.size() != d_slices.size() || ctxs.size() != d_tmps.size() || ctxs.size() != n_per_dev.size())
        throw std::invalid_argument("radix_sort_sharded: one context, slice, tmp and length per device");
    const rsx_layout L = RadixDigits<T>::layout();
    std::vector<rsx_ctx*> h;
    std::vector<void*> s, t;
    for (size_t g = 0; g < ctxs.size(); ++g) {
        h.push_back(ctxs[g]->get());
        s.push_back(d_slices[g]);
        t.push_back(d_tmps[g]);
    }
    ctxs[0]->check(rsx_sort_sharded(h.data(), (uint32_t)h.size(), s.data(), t.data(), n_per_dev.data(), &L),
                   "rsx_sort_sharded");
}

}  // namespace rsx
