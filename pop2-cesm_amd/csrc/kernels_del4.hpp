// kernels_del4.hpp -- biharmonic horizontal mixing (placeholder until the del4 kernels land)
#pragma once
namespace pop {
inline int del4_create(HostModel &, const DevGrid &, MixDev &, std::vector<void *> &, std::string &err) { err = "del4 horizontal mixing is not built yet"; return 1; }
inline int mix_hdifft_del4(const HostModel &, const DevGrid &, const StepParams &, const MixDev &, const double *, const double *, double *, double *, double *, double *, hipStream_t, std::string &err) { err = "del4 not built"; return 1; }
inline int mix_hdiffu_del4(const HostModel &, const DevGrid &, const StepParams &, const MixDev &, const double *, const double *, double *, double *, double *, double *, hipStream_t, std::string &err) { err = "del4 not built"; return 1; }
}  // namespace pop
