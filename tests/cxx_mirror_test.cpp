// Exercises the C++ host mirror (radix_sort_amd/cxx/radix_sort.hpp) the way the reference's
// tests exercise `radix_sort()` (src/radix_sort/tests.rs): random data per type, compared with
// the standard library's (stable) sort.  Run by tests/test_cxx_mirror.py on the GPU box.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "../radix_sort_amd/cxx/radix_sort.hpp"

template <typename T>
static bool bits_equal(const T& a, const T& b) { return std::memcmp(&a, &b, sizeof(T)) == 0; }

template <typename T, typename Gen>
static int test_ints(const char* name, Gen gen, size_t n) {
    std::vector<T> v(n);
    for (auto& x : v) x = gen();
    std::vector<T> exp = v;
    std::sort(exp.begin(), exp.end());  // tests.rs:7-23: == slice::sort()
    rsx::radix_sort(v);
    const bool ok = v == exp;
    std::printf("%-12s n=%zu %s\n", name, n, ok ? "ok" : "MISMATCH");
    return ok ? 0 : 1;
}

template <typename F, typename U>
static int test_float(const char* name, size_t n) {
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<F> d(-1, 1);
    std::vector<F> v(n);
    for (auto& x : v) x = d(rng);
    // tests.rs:139-143: 0.0, -0.0, NaN, +inf, -inf at random positions
    const F specials[] = {F(0.0), F(-0.0), std::numeric_limits<F>::quiet_NaN(), std::numeric_limits<F>::infinity(),
                          -std::numeric_limits<F>::infinity()};
    for (F s : specials) v[rng() % n] = s;
    std::vector<F> exp = v;
    auto key = [](F x) {  // total_cmp order == order of the mapped bit pattern
        U b;
        std::memcpy(&b, &x, sizeof b);
        const U top = U(1) << (sizeof(U) * 8 - 1);
        return (b & top) ? U(~b) : U(b | top);
    };
    std::stable_sort(exp.begin(), exp.end(), [&](F a, F b) { return key(a) < key(b); });
    rsx::radix_sort(v);
    bool ok = true;
    for (size_t i = 0; i < n; ++i) ok &= bits_equal(v[i], exp[i]);  // bitwise, tests.rs:146-151
    std::printf("%-12s n=%zu %s\n", name, n, ok ? "ok" : "MISMATCH");
    return ok ? 0 : 1;
}

int main() {
    const size_t n = 1000000;  // tests.rs: 1e6 elements per type
    std::mt19937_64 rng(1);
    int bad = 0;
    bad += test_ints<uint8_t>("u8", [&] { return (uint8_t)rng(); }, n);
    bad += test_ints<uint16_t>("u16", [&] { return (uint16_t)rng(); }, n);
    bad += test_ints<uint32_t>("u32", [&] { return (uint32_t)rng(); }, n);
    bad += test_ints<uint64_t>("u64", [&] { return (uint64_t)rng(); }, n);
    bad += test_ints<int8_t>("i8", [&] { return (int8_t)rng(); }, n);
    bad += test_ints<int16_t>("i16", [&] { return (int16_t)rng(); }, n);
    bad += test_ints<int32_t>("i32", [&] { return (int32_t)rng(); }, n);
    bad += test_ints<int64_t>("i64", [&] { return (int64_t)rng(); }, n);
    bad += test_ints<unsigned __int128>("u128", [&] { return ((unsigned __int128)rng() << 64) | rng(); }, n);
    bad += test_ints<__int128>("i128", [&] { return (__int128)(((unsigned __int128)rng() << 64) | rng()); }, n);
    bad += test_float<float, uint32_t>("f32", n);
    bad += test_float<double, uint64_t>("f64", n);
    {  // tests.rs:175-187: Vec<(u32,u32)>, both fields random, vs STABLE sort_by_key(.0), full equality
        std::vector<std::pair<uint32_t, uint32_t>> v(n);
        for (auto& x : v) x = {(uint32_t)(rng() & 0xFFFFF), (uint32_t)rng()};  // 2^20 keys: many duplicates
        auto exp = v;
        std::stable_sort(exp.begin(), exp.end(), [](auto& a, auto& b) { return a.first < b.first; });
        rsx::radix_sort(v);
        const bool ok = v == exp;
        std::printf("%-12s n=%zu %s\n", "(u32,u32)", n, ok ? "ok" : "MISMATCH");
        bad += !ok;
    }
    {  // RadixDigits::get_digit mirrors radix_digits.rs
        const bool ok = rsx::RadixDigits<uint32_t>::get_digit(0x11223344u, 2) == 0x22 &&
                        rsx::RadixDigits<int32_t>::get_digit(-1, 3) == 0x7F &&
                        rsx::RadixDigits<float>::get_digit(-1.0f, 3) == 0x40 &&
                        rsx::RadixDigits<std::pair<uint64_t, uint64_t>>::NUMBER_OF_DIGITS == 8;
        std::printf("%-12s %s\n", "get_digit", ok ? "ok" : "MISMATCH");
        bad += !ok;
    }
    std::vector<uint32_t> empty;
    rsx::radix_sort(empty);  // no panic, no-op
    std::printf(bad ? "FAILED %d\n" : "ALL OK\n", bad);
    return bad ? 1 : 0;
}
