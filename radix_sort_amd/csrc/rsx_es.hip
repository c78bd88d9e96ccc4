// rsx_es.hip -- the kernels and launchers of ONE element size: compiled once per size with
// -DRSX_ES=<bytes> (radix_sort_amd/_build.py), so that the eight sizes build in parallel and a
// tuning variant of one size relinks in seconds.
#ifndef RSX_ES
#error "compile with -DRSX_ES=<element bytes>"
#endif
#include "rsx_launch_impl.hpp"

namespace rsxh {
template int launch_hist<RSX_ES>(rsx_ctx*, const void*, const RegionGeom&, const rsx_layout*, uint32_t,
                                 unsigned long long*, unsigned long long*, bool, hipStream_t);
template int launch_hist2<RSX_ES>(rsx_ctx*, const void*, const RegionGeom&, const rsx_layout*, uint32_t, unsigned long long*,
                                  uint32_t, unsigned long long*, unsigned long long*, hipStream_t);
template int launch_wideplan<RSX_ES>(rsx_ctx*, const void*, size_t, const rsx_layout*, WidePlan*, hipStream_t);
template int launch_count16top<RSX_ES>(rsx_ctx*, const void*, size_t, const rsx_layout*, WidePlan*, uint32_t*, uint32_t, uint32_t, uint32_t, hipStream_t);
template int launch_marginal16<RSX_ES>(rsx_ctx*, const uint32_t*, uint32_t, uint32_t, const RegionGeom&, unsigned long long*, unsigned long long*, hipStream_t);
template int launch_bucket16<RSX_ES>(rsx_ctx*, void*, void*, size_t, const rsx_layout*, const uint64_t*, const WidePlan*, hipStream_t);
template int launch_mid_split<RSX_ES>(rsx_ctx*, const void*, void*, size_t, const rsx_layout*, hipStream_t);
template int launch_bucket_sort<RSX_ES>(rsx_ctx*, const void*, void*, const RegionGeom&, const rsx_layout*, hipStream_t);
template int launch_sweep<RSX_ES>(rsx_ctx*, const void*, void*, const RegionGeom&, const rsx_layout*, uint32_t,
                                  const unsigned long long*, unsigned long long*, unsigned long long*, int, hipStream_t);
template int launch_small_sort<RSX_ES>(rsx_ctx*, void*, size_t, const rsx_layout*, hipStream_t);
template int launch_segcopy<RSX_ES>(rsx_ctx*, const void*, void*, const uint64_t*, const uint64_t*, const uint64_t*,
                                    uint32_t, hipStream_t);
}  // namespace rsxh
