// bench_demo -- the caller of the path: the reference's bench protocol (src/main.rs:16-45,101-127)
// over the C++ host mirror, for a box without a Rust toolchain.
//
//   for (u32,u32) and (u64,u64): for size in 0.5, 1.0, ... 4.0 GB: `runs` times { fresh uniform keys
//   (payload 0, distr.rs KeyUniform), time ONLY the sort (main.rs:32-34) }, print the mean as
//   "Sorted {:.1}GB of data in: {:.4}s" (main.rs:113,124).
//
// Two timings per size: the literal drop-in (`radix_sort(&mut [T])` on a host vector: H2D + sort + D2H,
// PCIe-bound) and, with --device, the device-resident sort of the same number of elements.
// Also the reference's raw dataset files (headerless, native-endian array of T; main.rs:47-99):
//   bench_demo --gen-data <GB> <u32|u64> <file>      write one        (main.rs:82-99 gen_data)
//   bench_demo --data <u32|u64> <file> [file ...]    sort each, mean  (main.rs:47-80 bench_sorts_data)
//
// build: g++ -O2 -std=c++17 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include bench_demo.cpp ../lib/librsx.so \
//            -L/opt/rocm/lib -lamdhip64 -lpthread -Wl,-rpath,$PWD/../lib   (tests/test_cxx_mirror.py does exactly this)
#include <hip/hip_runtime_api.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "radix_sort.hpp"

namespace {

using Clock = std::chrono::steady_clock;

uint64_t mix(uint64_t x) {  // splitmix64
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// KeyUniform (distr.rs): uniform key, payload 0 -- filled by all host threads (not timed).  With
// `index_payload` (--check) the payload is the element's index instead, which makes instability visible.
template <typename K>
void fill_uniform(std::vector<std::pair<K, K>>& v, uint64_t seed, bool index_payload = false) {
    const unsigned nt = std::max(1u, std::thread::hardware_concurrency());
    std::vector<std::thread> th;
    const size_t n = v.size();
    for (unsigned t = 0; t < nt; ++t)
        th.emplace_back([&, t] {
            for (size_t i = n * t / nt; i < n * (t + 1) / nt; ++i) v[i] = {(K)mix(seed + i), index_payload ? (K)i : K(0)};
        });
    for (auto& x : th) x.join();
}

// order-independent checksum of the elements (the sort must keep the multiset)
template <typename K>
uint64_t multiset_sum(const std::vector<std::pair<K, K>>& v) {
    uint64_t s = 0;
    for (const auto& e : v) s += mix((uint64_t)e.first * 0x9E3779B97F4A7C15ull ^ (uint64_t)e.second);
    return s;
}

// ascending by key; equal keys in ascending payload (= original index under --check): the stable order
template <typename K>
bool is_sorted_stably(const std::vector<std::pair<K, K>>& v, bool payload_is_index) {
    for (size_t i = 1; i < v.size(); ++i) {
        if (v[i - 1].first > v[i].first) return false;
        if (payload_is_index && v[i - 1].first == v[i].first && v[i - 1].second > v[i].second) return false;
    }
    return true;
}

template <typename K>
size_t elems_for(double gb) {  // main.rs:24: size * 1e9 / size_of::<T>()
    return (size_t)((float)gb * 1e9f / (float)sizeof(std::pair<K, K>));
}

template <typename K>
double time_host(double gb, int runs, bool check) {
    const size_t n = elems_for<K>(gb);
    std::vector<std::pair<K, K>> data(n);
    double total = 0;
    for (int r = 0; r < runs; ++r) {
        fill_uniform(data, 0x5EED0000ull + (uint64_t)r * 0x100000000ull, check);
        const uint64_t before = check ? multiset_sum(data) : 0;
        const auto t0 = Clock::now();
        rsx::radix_sort(data);  // the drop-in: <[T]>::radix_sort(&mut self)
        total += std::chrono::duration<double>(Clock::now() - t0).count();
        if (check && (!is_sorted_stably(data, true) || multiset_sum(data) != before)) {
            std::fprintf(stderr, "NOT SORTED / NOT STABLE / ELEMENTS CHANGED\n");
            std::exit(2);
        }
    }
    return total / runs;
}

#define HIP_OK(x)                                                              \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) {                                                \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));       \
            std::exit(3);                                                      \
        }                                                                      \
    } while (0)

template <typename K>
double time_device(double gb, int runs) {
    using T = std::pair<K, K>;
    const size_t n = elems_for<K>(gb);
    T *d = nullptr, *tmp = nullptr;
    HIP_OK(hipMalloc(reinterpret_cast<void**>(&d), n * sizeof(T)));
    HIP_OK(hipMalloc(reinterpret_cast<void**>(&tmp), n * sizeof(T)));
    rsx::Context& ctx = rsx::default_context();
    const rsx_layout L = rsx::RadixDigits<T>::layout();
    double total = 0;
    for (int r = -1; r < runs; ++r) {  // run -1 warms the workspace up
        ctx.check(rsx_generate_device(ctx.get(), d, n, &L, RSX_GEN_UNIFORM, 0x5EED0000ull + (uint64_t)r, 0.0, 0, nullptr),
                  "rsx_generate_device");
        HIP_OK(hipDeviceSynchronize());
        const auto t0 = Clock::now();
        rsx::radix_sort_device(d, tmp, n);
        ctx.check(rsx_ctx_check(ctx.get(), nullptr), "rsx_ctx_check");  // synchronises
        if (r >= 0) total += std::chrono::duration<double>(Clock::now() - t0).count();
    }
    HIP_OK(hipFree(d));
    HIP_OK(hipFree(tmp));
    return total / runs;
}

template <typename K>
void ladder(const char* name, const std::vector<double>& sizes, int runs, bool device, bool check) {
    std::printf("\nTYPE: %s RUNS: %d\n", name, runs);  // main.rs:107,118
    for (double gb : sizes) {
        const double s = time_host<K>(gb, runs, check);
        std::printf("Sorted %.1fGB of data in: %.4fs", gb, s);  // main.rs:113,124
        if (device) std::printf("   (device-resident: %.4fs)", time_device<K>(gb, runs));
        std::printf("\n");
        std::fflush(stdout);
    }
}

template <typename K>
int gen_data(double gb, const char* path) {
    std::vector<std::pair<K, K>> v(elems_for<K>(gb));
    fill_uniform(v, 0xDA7Aull);
    FILE* f = std::fopen(path, "wb");
    if (!f) return 1;
    const size_t w = std::fwrite(v.data(), sizeof(v[0]), v.size(), f);
    std::fclose(f);
    return w == v.size() ? 0 : 1;
}

template <typename K>
int bench_files(int argc, char** argv, int first) {
    double total = 0;
    int files = 0;
    for (int i = first; i < argc; ++i) {
        FILE* f = std::fopen(argv[i], "rb");
        if (!f) {
            std::fprintf(stderr, "cannot open %s\n", argv[i]);
            return 1;
        }
        std::fseek(f, 0, SEEK_END);
        const long bytes = std::ftell(f);
        std::fseek(f, 0, SEEK_SET);
        std::vector<std::pair<K, K>> v((size_t)bytes / sizeof(std::pair<K, K>));  // main.rs:59: len / size_of::<T>()
        if (std::fread(v.data(), sizeof(v[0]), v.size(), f) != v.size()) return 1;
        std::fclose(f);
        const uint64_t before = multiset_sum(v);
        const auto t0 = Clock::now();
        rsx::radix_sort(v);
        total += std::chrono::duration<double>(Clock::now() - t0).count();
        ++files;
        if (!is_sorted_stably(v, false) || multiset_sum(v) != before) {
            std::fprintf(stderr, "NOT SORTED: %s\n", argv[i]);
            return 2;
        }
    }
    if (files) std::printf("Sorted %d file(s), mean: %.4fs\n", files, total / files);
    return 0;
}

}  // namespace

int main(int argc, char** argv) try {
    std::vector<double> sizes = {0.5, 1.0, 1.5, 2.0, 2.5, 3.0, 3.5, 4.0};  // main.rs:104
    int runs = 5;                                                           // main.rs:102
    bool device = false, check = false;
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "--runs" && i + 1 < argc) runs = std::atoi(argv[++i]);
        else if (a == "--device") device = true;
        else if (a == "--check") check = true;
        else if (a == "--sizes" && i + 1 < argc) {
            sizes.clear();
            for (char* tok = std::strtok(argv[++i], ","); tok; tok = std::strtok(nullptr, ",")) sizes.push_back(std::atof(tok));
        } else if (a == "--gen-data" && i + 3 < argc) {
            const double gb = std::atof(argv[i + 1]);
            return std::string(argv[i + 2]) == "u64" ? gen_data<uint64_t>(gb, argv[i + 3]) : gen_data<uint32_t>(gb, argv[i + 3]);
        } else if (a == "--data" && i + 2 < argc) {
            return std::string(argv[i + 1]) == "u64" ? bench_files<uint64_t>(argc, argv, i + 2) : bench_files<uint32_t>(argc, argv, i + 2);
        } else {
            std::fprintf(stderr, "usage: bench_demo [--runs N] [--sizes a,b,...] [--device] [--check] | --gen-data GB u32|u64 FILE | --data u32|u64 FILE...\n");
            return 64;
        }
    }
    if (runs <= 0 || sizes.empty()) return 0;  // main.rs:21-23
    ladder<uint32_t>("u32/u32", sizes, runs, device, check);
    ladder<uint64_t>("u64/u64", sizes, runs, device, check);
    return 0;
} catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
}
